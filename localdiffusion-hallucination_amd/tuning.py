"""Tuning of the build: every performance knob of the package in ONE explicit object.

``Tuning`` holds the host-side scheduling choices (concurrent sub-batches, host pacing, which fusions the plan
builder uses) and, under ``kernel``, overrides of the library's launch-routing table (``ld_tuning_set``,
include/localdiff_hip.h).  ``Unet(..., tuning=Tuning(...))`` / ``GaussianDiffusion.tuning`` take one; the default is
``Tuning.from_env()``: the compiled-in defaults (the values DESIGN.md's measurements settled on) with ``LD_*``
environment variables as overrides, parsed HERE and nowhere else in the package.  No knob changes what is computed --
only which kernel variant computes it and how launches are scheduled (results can differ in the summation order of
a tile variant, i.e. in the last bits).  The reference has no counterpart (its tuning is cuDNN's and ATen's).
"""
import os
from dataclasses import dataclass, field, fields, replace
from typing import Dict, Mapping, Optional, Tuple

_PROCESS_ENV = os.environ        # the package's ONE handle on the process environment


@dataclass
class Tuning:
    # ---- sampler scheduling (diffusion.py)
    sub_batches: int = 2                 # concurrent sub-batches of the joint / branch steps (1 = one batch, one stream)
    min_sub_batch: int = 2               # smallest sub-batch worth its own stream (in 256^2-patch equivalents)
    sub_resync: int = 32                 # steps between phase alignments of the sub-batch streams (0 = never)
    sub_resync_early: int = 1            # ... and in front of each of the first steps
    sub_ahead: int = 2                   # the host enqueues at most this many replayed steps ahead of the GPU (0 = no limit)
    sub_joint_graph: bool = False        # ONE captured graph per step that forks to every sub-batch stream and joins (one host launch per step)
    fused_final_step: bool = True        # final_conv + posterior update + noise draw as one launch
    fused_step_begin: bool = True        # the head-of-step work (arena reset, step counter, FiLM row) inside init_conv's launch (16-bit storage)
    # ---- plan builder (unet.py)
    weight_split_levels: int = 0         # two-term (hi + lo) weights on the first N resolution levels (accuracy mode)
    separate_act: bool = True            # block1's GroupNorm + FiLM + SiLU as its own pass on small, wide maps
    sep_act_max_px: int = 32 * 32        # ... maps of at most this many pixels
    sep_act_min_c: int = 128             # ... with at least this many channels
    fusion_fold: bool = True             # conv_fusion's conditioning halves evaluated once per sample (16-bit storage)
    side_res_conv: bool = True           # a ResnetBlock's res_conv rides in its block1 convolution's launch (16-bit storage); the tail is a gn_apply
    side_res_conv_max_px: int = 64 * 64  # ... on maps of at most this many pixels (measured: 32^2 and 64^2 win; at 128^2 the launch leaves the 64 x 16-row tile and loses, at 256^2 the 16-row tile has no registers for the side accumulators)
    side_res_conv_px: Optional[Tuple[int, ...]] = None   # experiments: exactly these map sizes (pixels) instead of the max_px rule
    linattn_chunk_px: Optional[Tuple[int, int, int]] = None   # kvctx chunk pixels for n >= 65536 / n >= 16384 / smaller (None: by batch)
    graph_prewarm: bool = True           # a captured graph is launched once where it is captured (the first launch completes its set-up)
    pool_by_size: bool = True            # pool placement: largest buffers first (False: by first use; 9 % more pool at the bench shape)
    recompute_stem: bool = False         # pooled plans: init_conv evaluated a second time for the final block's concat instead of kept live
    buffer_reuse: bool = True            # sampler plans (table mode): activations share ONE pool by liveness instead of a buffer per layer
    pool_verify: bool = False            # debugging: every pooled buffer is overwritten with 0xFF bytes (NaNs) right behind the launch of its last DECLARED use
    # ---- the library's launch-routing table (ld_tuning_set): name -> value, applied when the library is loaded
    kernel: Dict[str, int] = field(default_factory=dict)

    # environment variable -> (field, parser); LD_NO_* switches turn a default-on feature off
    _ENV = {
        "LD_SUB_BATCHES": ("sub_batches", int), "LD_MIN_SUB_BATCH": ("min_sub_batch", int),
        "LD_SUB_RESYNC": ("sub_resync", int), "LD_SUB_RESYNC_EARLY": ("sub_resync_early", int),
        "LD_SUB_AHEAD": ("sub_ahead", int), "LD_SUB_JOINT_GRAPH": ("sub_joint_graph", lambda v: v not in ("0", "")), "LD_WEIGHT_SPLIT_LEVELS": ("weight_split_levels", int),
        "LD_SEP_ACT_MAX_PX": ("sep_act_max_px", int), "LD_SEP_ACT_MIN_C": ("sep_act_min_c", int),
        "LD_NO_FUSED_FINAL": ("fused_final_step", lambda v: False), "LD_NO_FUSED_BEGIN": ("fused_step_begin", lambda v: False),
        "LD_NO_SEPARATE_ACT": ("separate_act", lambda v: False),
        "LD_NO_FUSION_FOLD": ("fusion_fold", lambda v: False),
        "LD_NO_SIDE_RES_CONV": ("side_res_conv", lambda v: False), "LD_SIDE_RES_CONV_MAX_PX": ("side_res_conv_max_px", int),
        "LD_SIDE_RES_CONV_PX": ("side_res_conv_px", lambda v: tuple(int(x) for x in v.split(","))),
        "LD_NO_GRAPH_PREWARM": ("graph_prewarm", lambda v: False),
        "LD_POOL_BY_START": ("pool_by_size", lambda v: False),
        "LD_RECOMPUTE_STEM": ("recompute_stem", lambda v: v not in ("0", "")),
        "LD_BUFFER_REUSE": ("buffer_reuse", lambda v: v not in ("0", "")), "LD_NO_BUFFER_REUSE": ("buffer_reuse", lambda v: False),
        "LD_POOL_VERIFY": ("pool_verify", lambda v: v not in ("0", "")),
        "LD_LINATTN_CHUNK_PX": ("linattn_chunk_px", lambda v: tuple(int(x) for x in (v.split(",") * 3)[:3])),
    }

    @classmethod
    def from_env(cls, env: Optional[Mapping[str, str]] = None, **overrides) -> "Tuning":
        """Defaults, then ``LD_*`` overrides from ``env`` (default: the process environment), then ``overrides``.
        The kernel-side table reads its own ``LD_<NAME>`` overrides inside the library (runtime.hip), once."""
        env = _PROCESS_ENV if env is None else env
        t = cls()
        for var, (name, parse) in cls._ENV.items():
            if env.get(var) not in (None, ""):
                setattr(t, name, parse(env[var]))
        return replace(t, **overrides) if overrides else t

    def chunk_rule(self, batch: int) -> Tuple[int, int, int]:
        """kvctx pixels per chunk for maps of n >= 65536 / n >= 16384 / fewer pixels (DESIGN findings 36, 68)."""
        if self.linattn_chunk_px is not None:
            return tuple(self.linattn_chunk_px)
        return (512, 256, 128) if batch <= 8 else (1024, 256, 128)

    def apply_kernel_table(self, lib) -> None:
        """Write ``kernel`` into the loaded library's routing table (unknown names raise through the C ABI)."""
        from . import _cabi as cabi
        for name, value in self.kernel.items():
            cabi.check(lib.ld_tuning_set(name.encode(), int(value)), f"tuning {name}")

    def describe(self) -> dict:
        return {f.name: getattr(self, f.name) for f in fields(self) if not f.name.startswith("_")}


def kernel_table(lib) -> Dict[str, int]:
    """The library's routing table as it stands (defaults + environment + explicit sets)."""
    import ctypes as C
    out = {}
    for i in range(lib.ld_tuning_count()):
        name = lib.ld_tuning_name(i)
        v = C.c_longlong()
        lib.ld_tuning_get(name, C.byref(v))
        out[name.decode()] = int(v.value)
    return out


def lib_override_path() -> Optional[str]:
    """A/B builds: another liblocaldiff_hip.so to load instead of the in-tree one (LD_LIB_OVERRIDE)."""
    return _PROCESS_ENV.get("LD_LIB_OVERRIDE") or None


def runtime_env_default(var: str, value: str) -> bool:
    """Set a HIP-runtime environment default unless the caller's environment already decides it."""
    if var in _PROCESS_ENV:
        return False
    _PROCESS_ENV[var] = value
    return True


def rendezvous_run_id() -> Optional[str]:
    """What distinguishes this run's rendezvous files from another run's (dist.LdComm.bootstrap)."""
    return _PROCESS_ENV.get("TORCHELASTIC_RUN_ID") or _PROCESS_ENV.get("MASTER_PORT")


RUNTIME_CONFIGURED = False      # set by configure_runtime(), whatever it was asked to apply


def runtime_configured(var: str = "DEBUG_CLR_GRAPH_PACKET_CAPTURE") -> bool:
    """Did the caller decide the runtime settings -- by calling ``configure_runtime`` (with any arguments) or by setting
    the variable in the environment?"""
    return RUNTIME_CONFIGURED or var in _PROCESS_ENV
