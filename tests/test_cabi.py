"""CPU-side checks of the drop-in boundary: the shared library loads, exports every function that
include/localdiff_hip.h declares (and the ctypes table lists exactly those), rejects bad arguments
without touching a GPU, and the product path refuses to run without one."""
import ctypes as C
import os
import re

import pytest
import torch

import localdiffusion_hallucination_amd as ldh
from localdiffusion_hallucination_amd import _cabi as cabi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "localdiff_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ld_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_function():
    names = header_functions()
    assert len(names) >= 35
    lib = cabi.lib()
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
    assert sorted(cabi.EXPORTS) == names, set(cabi.EXPORTS) ^ set(names)
    assert lib.ld_version() >= 100


def test_struct_layouts_match_header_field_order():
    src = open(os.path.join(ROOT, "include", "localdiff_hip.h")).read()
    for cname, pstruct in [("ld_src", cabi.Src), ("ld_conv3x3_args", cabi.Conv3x3Args),
                           ("ld_conv1x1_args", cabi.Conv1x1Args), ("ld_gn_apply_args", cabi.GnApplyArgs)]:
        body = re.search(r"typedef struct " + cname + r" \{(.*?)\} " + cname + ";", src, flags=re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            names = decl.split(",")
            first = re.findall(r"([A-Za-z_][A-Za-z0-9_]*)\s*(?:\[\d+\])?$", names[0].strip())[0]
            fields.append(first)
            fields += [n.strip() for n in names[1:]]
        assert [f[0] for f in pstruct._fields_] == fields, (cname, fields)


def test_argument_validation_needs_no_gpu():
    lib = cabi.lib()
    a = cabi.Conv3x3Args()
    a.nsrc = 3
    assert lib.ld_conv3x3(C.byref(a), None) == -1
    assert b"nsrc" in lib.ld_last_error()
    a.nsrc, a.dtype, a.Cout = 1, 7, 32
    assert lib.ld_conv3x3(C.byref(a), None) == -1 and b"dtype" in lib.ld_last_error()
    assert lib.ld_attention(None, None, 1, 4, 4, 32, 0, None) == -1
    assert lib.ld_linattn_kvctx(None, None, None, None, 1, 4, 32, 4, 32, 1, 1, None) == -1
    assert int(lib.ld_linattn_ctx_part_floats(2, 4, 32, 3)) == 2 * 4 * 3 * (1024 + 64)
    with pytest.raises(RuntimeError):
        cabi.check(-1, "demo")


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_product_path_fails_loudly_without_gpu():
    net = ldh.Unet(dim=32, init_dim=32, dim_mults=(1, 2, 4), full_attn=(False, False, True), mode="mnist")
    with pytest.raises(RuntimeError, match="GPU"):
        net(torch.zeros(1, 1, 28, 28), torch.zeros(1, 1, 28, 28), torch.zeros(1, dtype=torch.long))
    cfg = dict(branch_out=False, start_intermediate=False, start_timestep=2, data="mnist", mask_x=False)
    gd = ldh.GaussianDiffusion(cfg, net, image_size=28, timesteps=4, objective="pred_x0")
    with pytest.raises(RuntimeError):
        gd.sample(torch.zeros(1, 1, 28, 28), None, batch_size=1, min_max_val=(0.0, 2.0))


def test_state_dict_round_trip_and_reference_names():
    from localdiffusion_hallucination_amd import weights
    net = ldh.Unet(dim=32, init_dim=32, channels=3, out_dim=3, mode="mvtec")
    assert list(net.state_dict().keys()) == list(weights.unet_param_shapes(net.cfg).keys())
    cfg = dict(branch_out=True, start_intermediate=True, start_timestep=2, data="mvtec", mask_x=True)
    gd = ldh.GaussianDiffusion(cfg, net, image_size=64, timesteps=50, objective="pred_x0", sampling_timesteps=10)
    keys = list(gd.state_dict().keys())
    assert keys[:13] == list(ldh.schedule.BUFFER_NAMES) and keys[13] == "model.cond_model.residual_conv1.0.convblock.0.weight"
    assert len(keys) == 13 + 334
    assert gd.is_ddim_sampling and gd.num_timesteps == 50 and gd.channels == 3
    sd = {k: torch.from_numpy(v) for k, v in weights.procedural_state_dict(net.cfg, 1).items()}
    net.load_state_dict(sd)
    assert torch.equal(net.state_dict()["final_conv.bias"], sd["final_conv.bias"])


def test_rccl_entry_points_load_lazily():
    """ld_comm_* / ld_allgather dlopen RCCL on first use: the library itself loads without it (checked by every other
    test here), a unique id can be made on a box without a GPU, and bad arguments are refused before RCCL is touched."""
    import ctypes as C
    lib = cabi.lib()
    buf = (C.c_char * 128)()
    rc = lib.ld_comm_unique_id(buf)
    assert rc == 0 or b"RCCL unavailable" in lib.ld_last_error()
    comm = C.c_void_p()
    assert lib.ld_comm_init(C.byref(comm), buf, 2, 5) == -1 and b"bad arguments" in lib.ld_last_error()
    assert lib.ld_allgather(None, None, 0, None, None) == -1
