"""The build's tuning lives in ONE object per side: ``Tuning`` for the host-side scheduling and plan
choices, the library's launch-routing table behind ``ld_tuning_*``.  Defaults are pinned here -- they are the values the
measurements in docs/findings.md settled on -- and the environment is only an override, parsed in one place."""
import ctypes as C
import glob
import os
import re

import pytest

import localdiffusion_hallucination_amd as ldh
from localdiffusion_hallucination_amd import _cabi as cabi
from localdiffusion_hallucination_amd.tuning import Tuning, kernel_table

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_defaults_are_the_measured_ones():
    t = Tuning.from_env(env={})
    assert t.describe() == dict(sub_batches=2, min_sub_batch=2, sub_resync=32, sub_resync_early=1, sub_ahead=2, sub_joint_graph=False,
                                fused_final_step=True, fused_step_begin=True, weight_split_levels=0, separate_act=True, sep_act_max_px=1024,
                                sep_act_min_c=128, fusion_fold=True, side_res_conv=True, side_res_conv_max_px=4096, side_res_conv_px=None, linattn_chunk_px=None, graph_prewarm=True, pool_by_size=True, recompute_stem=False, buffer_reuse=True, pool_verify=False, kernel={})
    assert t.chunk_rule(4) == (512, 256, 128) and t.chunk_rule(8) == (512, 256, 128) and t.chunk_rule(32) == (1024, 256, 128)


def test_environment_is_an_override_parsed_in_one_place():
    t = Tuning.from_env(env={"LD_SUB_BATCHES": "1", "LD_NO_SEPARATE_ACT": "1", "LD_LINATTN_CHUNK_PX": "256", "LD_SUB_AHEAD": "0",
                             "LD_NO_FUSED_FINAL": "1", "LD_WEIGHT_SPLIT_LEVELS": "2", "LD_UNRELATED": "7"})
    assert (t.sub_batches, t.separate_act, t.linattn_chunk_px, t.sub_ahead, t.fused_final_step, t.weight_split_levels) == \
        (1, False, (256, 256, 256), 0, False, 2)
    assert Tuning.from_env(env={}, sub_batches=4).sub_batches == 4
    # no other module of the package touches the process environment
    offenders = []
    for f in glob.glob(os.path.join(ROOT, "localdiffusion-hallucination_amd", "*.py")):
        if os.path.basename(f) != "tuning.py" and re.search(r"os\.environ|getenv", open(f).read()):
            offenders.append(os.path.basename(f))
    assert offenders == []
    assert len(re.findall(r"os\.environ", open(os.path.join(ROOT, "localdiffusion-hallucination_amd", "tuning.py")).read())) == 1


def test_objects_carry_their_tuning():
    kw = dict(dim_mults=(1, 2, 4), full_attn=(False, False, True), mode="mnist")
    net = ldh.Unet(dim=32, init_dim=32, tuning=Tuning(sub_batches=1, weight_split_levels=2, separate_act=False), **kw)
    assert net.weight_split_levels == 2 and net.tuning.separate_act is False
    cfg = dict(branch_out=False, start_intermediate=False, start_timestep=2, data="mnist", mask_x=False)
    gd = ldh.GaussianDiffusion(cfg, net, image_size=28, timesteps=10, objective="pred_x0")
    assert gd.tuning is net.tuning and gd.sub_batches == 1 and gd.fuse_final_step is True


def test_kernel_table_defaults_and_explicit_sets():
    """The library's routing table through the C ABI (no GPU needed): defaults, set / get, unknown names refused."""
    lib = cabi.lib()
    names = [lib.ld_tuning_name(i).decode() for i in range(lib.ld_tuning_count())]
    assert lib.ld_tuning_name(lib.ld_tuning_count()) is None
    defaults = dict(c1_group=1, c1_group_max_px=32768, c1_group_min_ch=4, c1_pair_max_px=1 << 40, c1_small_min=256, conv_raw=1,
                    conv_mt4_min_wgs=256, conv_big_min=512, conv_sk=0, conv_sk_max_wgs=256, conv_c32=0, conv_c32_min_tiles=2048, conv_s32=3, conv_s32_min_tiles=1024, conv_big4_min=256,
                    gn_frags_per_block=512, fold_split_min=32, attn_split_max_wgs=256, attn_split_min_n=2048, lead_args=1, attn_xcd_map=1)
    assert sorted(names) == sorted(defaults)
    overridden = {n for n in names if os.environ.get("LD_" + n.upper()) is not None}
    table = kernel_table(lib)
    assert {k: v for k, v in table.items() if k not in overridden} == {k: v for k, v in defaults.items() if k not in overridden}
    try:
        Tuning(kernel={"conv_c32_min_tiles": 1024}).apply_kernel_table(lib)
        assert kernel_table(lib)["conv_c32_min_tiles"] == 1024
    finally:
        lib.ld_tuning_set(b"conv_c32_min_tiles", table["conv_c32_min_tiles"])
    v = C.c_longlong()
    assert lib.ld_tuning_get(b"no_such_entry", C.byref(v)) != 0 and lib.ld_tuning_set(b"no_such_entry", 1) != 0
    with pytest.raises(RuntimeError):
        Tuning(kernel={"no_such_entry": 1}).apply_kernel_table(lib)


def test_only_runtime_hip_reads_tuning_from_the_environment():
    """csrc/: getenv appears in runtime.hip (the table, roctx opt-in), collective.hip (LD_RCCL_PATH: a path, not a
    threshold) and inside LD_DEBUG_VARIANTS blocks (experiment switches of --debug-variants builds) -- nowhere else."""
    bad = []
    for f in sorted(glob.glob(os.path.join(ROOT, "localdiffusion-hallucination_amd", "csrc", "*.hip")) +
                    glob.glob(os.path.join(ROOT, "localdiffusion-hallucination_amd", "csrc", "*.h"))):
        name = os.path.basename(f)
        if name in ("runtime.hip", "collective.hip"):
            continue
        stack = []                      # per open #if: "dbg" (inside #ifdef LD_DEBUG_VARIANTS), "prod" (its #else arm), "other"
        for ln, line in enumerate(open(f), 1):
            s = line.strip()
            if s.startswith("#if"):
                stack.append("dbg" if s.startswith("#ifdef LD_DEBUG_VARIANTS") else "other")
            elif s.startswith("#else") and stack and stack[-1] in ("dbg", "prod"):
                stack[-1] = "prod" if stack[-1] == "dbg" else "dbg"
            elif s.startswith("#endif") and stack:
                stack.pop()
            if "getenv(" in line and not s.startswith("//") and "dbg" not in stack:
                bad.append(f"{name}:{ln}")
    assert bad == []
