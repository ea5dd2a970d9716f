"""ctypes binding of ``csrc/liblocaldiff_hip.so`` (declared in ``include/localdiff_hip.h``).

This is the only way the package reaches the GPU: there is no CPU fallback.  ``lib()`` raises
``RuntimeError`` when the shared library is missing or a symbol the header declares cannot be
resolved, so a broken build can never silently run something else.
"""
import ctypes as C
import os

from .tuning import lib_override_path

_HERE = os.path.dirname(os.path.abspath(__file__))

LIB_PATH = lib_override_path() or os.path.join(_HERE, "csrc", "liblocaldiff_hip.so")   # override: A/B builds

LD_F32, LD_BF16, LD_F16 = 0, 1, 2
ACT_NONE, ACT_SILU, ACT_RELU = 0, 1, 2
EPI_PLAIN, EPI_QKV_LINEAR, EPI_QKV_FULL, EPI_RMS_RES, EPI_RES, EPI_GN_TAIL = 0, 1, 2, 3, 4, 5
OBJ = {"pred_x0": 0, "pred_noise": 1, "pred_v": 2}
SCHED_COLS = 8
STAT_STRIPES = 16      # LD_STAT_STRIPES
COUNTER_CONV3X3_C32, COUNTER_CONV3X3_GENERIC, COUNTER_CONV3X3_S32 = 0, 1, 2

vp, i32, i64, u64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_float


class Src(C.Structure):
    _fields_ = [("data", vp), ("C", i32), ("pix_stride", i32), ("upsample", i32), ("gn_stats", vp),
                ("gn_gamma", vp), ("gn_beta", vp), ("gn_groups", i32), ("act", i32), ("film", vp),
                ("film_tstride", i32), ("film_bstride", i32)]


class Conv3x3Args(C.Structure):
    _fields_ = [("src", Src * 2), ("nsrc", i32), ("weight", vp), ("bias", vp), ("out", vp),
                ("out_stats", vp), ("out_groups", i32), ("B", i32), ("H", i32), ("W", i32),
                ("Cout", i32), ("t_ptr", vp), ("dtype", i32), ("addend", vp), ("weight_terms", i32),
                ("side_weight", vp), ("side_bias", vp), ("side_out", vp)]


class Conv1x1Args(C.Structure):
    _fields_ = [("src", Src * 2), ("nsrc", i32), ("unshuffle", i32), ("rms_in", i32), ("weight", vp),
                ("weight_bstride", i64), ("bias", vp), ("epilogue", i32), ("hidden", i32),
                ("q_scale", f32), ("g2", vp), ("residual", vp), ("gn_tail", Src), ("out", vp), ("kmax_out", vp), ("B", i32), ("H", i32),
                ("W", i32), ("Cout", i32), ("dtype", i32), ("weight_terms", i32)]


class GnApplyArgs(C.Structure):
    _fields_ = [("a", Src), ("b", Src), ("final_act", i32), ("pool", i32), ("out", vp), ("B", i32),
                ("H", i32), ("W", i32), ("t_ptr", vp), ("dtype", i32)]


class StepBeginArgs(C.Structure):
    _fields_ = [("zero_a", vp), ("bytes_a", C.c_size_t), ("zero_b", vp), ("bytes_b", C.c_size_t), ("t_ptr", vp), ("delta", i32),
                ("idx_ptr", vp), ("t_table", vp), ("film_rows", vp), ("row_floats", i32), ("film_cur", vp)]


# name -> (restype, argtypes); must list every function include/localdiff_hip.h declares
_SIGS = {
    "ld_last_error": (C.c_char_p, []),
    "ld_version": (C.c_int, []),
    "ld_device_info": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(i64)]),
    "ld_graph_begin": (C.c_int, [vp]),
    "ld_graph_end": (C.c_int, [vp, C.POINTER(vp)]),
    "ld_graph_launch": (C.c_int, [vp, vp]),
    "ld_graph_destroy": (C.c_int, [vp]),
    "ld_memset_zero": (C.c_int, [vp, C.c_size_t, vp]),
    "ld_memset_bytes": (C.c_int, [vp, C.c_int, C.c_size_t, vp]),
    "ld_event_create": (C.c_int, [C.POINTER(vp)]),
    "ld_event_record": (C.c_int, [vp, vp]),
    "ld_event_elapsed_ms": (C.c_int, [vp, vp, C.POINTER(f32)]),
    "ld_event_destroy": (C.c_int, [vp]),
    "ld_stream_wait_event": (C.c_int, [vp, vp]),
    "ld_counter": (C.c_longlong, [C.c_int]),
    "ld_tuning_set": (C.c_int, [C.c_char_p, C.c_longlong]),
    "ld_tuning_get": (C.c_int, [C.c_char_p, C.POINTER(C.c_longlong)]),
    "ld_tuning_count": (C.c_int, []),
    "ld_tuning_name": (C.c_char_p, [C.c_int]),
    "ld_range_push": (C.c_int, [C.c_char_p]),
    "ld_range_pop": (C.c_int, []),
    "ld_conv3x3": (C.c_int, [C.POINTER(Conv3x3Args), vp]),
    "ld_conv1x1": (C.c_int, [C.POINTER(Conv1x1Args), vp]),
    "ld_pack_conv_weight": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "ld_pack_conv_weight_terms": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "ld_conv_image": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                C.c_int, C.c_int, vp]),
    "ld_stem_packed_bytes": (C.c_size_t, []),
    "ld_pack_stem_weight": (C.c_int, [vp, vp, C.c_int, vp]),
    "ld_conv_stem": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "ld_conv_stem_begin": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(StepBeginArgs), vp]),
    "ld_gn_apply": (C.c_int, [C.POINTER(GnApplyArgs), vp]),
    "ld_linattn_kmax": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "ld_linattn_ctx": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "ld_linattn_ctx_reduce": (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, vp]),
    "ld_linattn_fold": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "ld_linattn_ctxfold": (C.c_int, [vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "ld_linattn_kvctx": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "ld_linattn_out": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, f32, C.c_int, vp]),
    "ld_linattn_kvctx_terms": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "ld_linattn_out_terms": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, f32, C.c_int, C.c_int, vp]),
    "ld_linattn_ctx_part_floats": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "ld_attention": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "ld_time_mlp": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, vp, vp, vp, C.c_int, vp, vp]),
    "ld_time_mlp_fourier": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, vp, vp, vp, C.c_int, vp, vp]),
    "ld_film": (C.c_int, [vp, C.c_int, C.c_int, vp, vp, C.c_int, vp, vp]),
    "ld_final_conv": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "ld_final_step": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, f32, f32, C.c_int, u64, i64, C.c_int, C.c_int, C.c_int,
                                C.c_int, C.c_int, C.c_int, vp]),
    "ld_final_step_at": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, f32, f32, C.c_int, u64, i64, i64, i64, vp, C.c_int,
                                   C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "ld_timing_begin": (C.c_int, [C.c_int]),
    "ld_timing_count": (C.c_int, []),
    "ld_timing_end": (C.c_int, [vp, C.c_int, vp]),
    "ld_timing_end_abs": (C.c_int, [vp, vp, C.c_int, vp]),
    "ld_randn": (C.c_int, [vp, i64, u64, i64, i64, vp, vp]),
    "ld_randn_at": (C.c_int, [vp, i64, i64, u64, i64, i64, vp, vp]),
    "ld_step_add": (C.c_int, [vp, C.c_int, vp]),
    "ld_step_begin": (C.c_int, [vp, C.c_size_t, vp, C.c_size_t, vp, C.c_int, vp, vp, vp]),
    "ld_step_begin_film": (C.c_int, [vp, C.c_size_t, vp, C.c_size_t, vp, C.c_int, vp, vp, vp, C.c_int, vp, vp]),
    "ld_ddim_step_at": (C.c_int, [vp, vp, vp, vp, vp, vp, f32, f32, C.c_int, i64, vp]),
    "ld_ddpm_step": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, f32, f32, C.c_int, i64, vp]),
    "ld_posterior_step": (C.c_int, [vp, vp, vp, vp, vp, vp, i64, vp]),
    "ld_ddim_step": (C.c_int, [vp, vp, vp, vp, f32, f32, f32, f32, f32, f32, f32, f32, f32, C.c_int,
                               C.c_int, i64, vp]),
    "ld_branch_conditions": (C.c_int, [vp, vp, vp, vp, f32, C.c_int, C.c_int, C.c_int, vp]),
    "ld_mask_out": (C.c_int, [vp, vp, f32, C.c_int, C.c_int, C.c_int, vp]),
    "ld_fuse_ddpm": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, f32, f32, C.c_int, C.c_int, C.c_int, vp]),
    "ld_branch_conditions_k": (C.c_int, [vp, vp, vp, f32, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "ld_fuse_ddpm_k": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, f32, f32, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "ld_fuse_ddim": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, f32, f32, f32, f32, f32, f32, f32, C.c_int,
                               C.c_int, C.c_int, vp]),
    "ld_fuse_ddim_k": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, f32, f32, f32, f32, f32, f32, f32, C.c_int, C.c_int,
                                 C.c_int, C.c_int, vp]),
    "ld_q_sample": (C.c_int, [vp, vp, vp, f32, f32, i64, vp]),
    "ld_q_sample_t": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, i64, vp]),
    "ld_p_losses": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, i64, C.c_int, vp]),
    "ld_recompose": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "ld_comm_unique_id": (C.c_int, [vp]),
    "ld_comm_init": (C.c_int, [C.POINTER(vp), vp, C.c_int, C.c_int]),
    "ld_comm_init_timeout": (C.c_int, [C.POINTER(vp), vp, C.c_int, C.c_int, C.c_double]),
    "ld_allgather": (C.c_int, [vp, vp, C.c_size_t, vp, vp]),
    "ld_comm_destroy": (C.c_int, [vp]),
}

EXPORTS = tuple(_SIGS)
_lib = None


def lib():
    """The loaded library (loads on first call).  Raises RuntimeError if it cannot be used."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or csrc/build.sh).  There is no CPU fallback for the HIP hot path.")
    try:
        l = C.CDLL(LIB_PATH)
    except OSError as e:
        raise RuntimeError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in _SIGS.items():
        try:
            fn = getattr(l, name)
        except AttributeError as e:
            raise RuntimeError(f"{LIB_PATH} does not export {name}") from e
        fn.restype, fn.argtypes = res, args
    _lib = l
    return l


LD_ETIMEOUT = -3


def check(rc, what=""):
    if rc != 0:
        msg = lib().ld_last_error().decode("utf-8", "replace")
        if rc == LD_ETIMEOUT:
            raise TimeoutError(f"localdiff_hip {what} timed out: {msg}")
        raise RuntimeError(f"localdiff_hip {what} failed (rc={rc}): {msg}")


class prof_range:
    """``with prof_range("step"):`` -- a roctx range around a phase of the sampler (ld_range_push / ld_range_pop)."""

    def __init__(self, name):
        self.name = name.encode()

    def __enter__(self):
        lib().ld_range_push(self.name)

    def __exit__(self, *exc):
        lib().ld_range_pop()


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else t.data_ptr()


def dtype_code(name):
    return {"fp32": LD_F32, "float32": LD_F32, "bf16": LD_BF16, "bfloat16": LD_BF16, "fp16": LD_F16, "float16": LD_F16,
            "half": LD_F16}[name]
