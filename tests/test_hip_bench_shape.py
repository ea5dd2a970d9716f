"""The benchmarked routing, in the benchmarked storage types, against the oracle (VERDICT r3 "what's weak" 1).

bench.py times cfg3 -- 8 patches of 3x256x256 -- as two concurrent sub-batches of FOUR patches (generic conv3x3 on the
256^2 maps, 512-pixel linear-attention chunks over n = 65,536) and prices its solo leg on one batch of EIGHT (the
persistent conv3x3_c32 takes the 32->32 @256^2 convolutions there).  Every other 16-bit comparison against the oracle
in tests/ is at <= 64^2; the ones at 256^2 compare the HIP path with itself.  Here one forward at B = 4 and B = 8 and a
short chain through the sub-batch runner are compared with oracle/unet_ref.py / oracle/diffusion_ref.py directly
(/root/reference/ddpm.py:404-451, 929-977), with absolute bounds written below."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import localdiffusion_hallucination_amd as ldh                   # noqa: E402
from localdiffusion_hallucination_amd import _cabi as cabi       # noqa: E402
from localdiffusion_hallucination_amd import rng, weights        # noqa: E402
from oracle import diffusion_ref, unet_ref                        # noqa: E402

H = 256
KW = dict(channels=3, out_dim=3, mode="mvtec")
# max-abs of ONE forward's output relative to the oracle output's max-abs (the bounds of test_forward_16bit_within_tolerance
# and smoke(): bf16 has 8 mantissa bits, fp16 11), and per recorded tap
OUT_TOL = {"bf16": 2e-2, "fp16": 3e-3}       # measured round 4: 6.7e-3 / 7.5e-3 (B = 4 / 8) and 9.1e-4 / 8.6e-4
TAP_TOL = {"bf16": 4e-2, "fp16": 5e-3}       # measured: worst tap 1.5e-2 (mid_block1) and 1.75e-3
# mean-abs / max-abs of x after T = 6 chained steps on the [0, 2] image range
# (measured round 4: bf16 7.3e-3 / 7.8e-2, fp16 8.5e-4 / 8.7e-3)
CHAIN_MEAN = {"bf16": 1.2e-2, "fp16": 1.5e-3}
CHAIN_MAX = {"bf16": 0.16, "fp16": 2e-2}


def _net(dtype):
    net = ldh.Unet(dim=32, init_dim=32, compute_dtype=dtype, **KW)
    sd = {k: torch.from_numpy(v) for k, v in weights.procedural_state_dict(net.cfg, 0).items()}
    net.load_state_dict(sd)
    return net.to("cuda"), sd


def _patch_conditions(B):
    """bench.py's per-patch conditioning (ddpm.py:677-688): band masks, patch 0 hard-masked, the others floored at 0.95."""
    img = torch.from_numpy(rng.uniform((1, 3, H, H), 100, 1, 0.0, 2.0))
    out = []
    for k in range(B):
        m = torch.zeros(1, 1, H, H)
        m[..., k * (H // 8):(k + 1) * (H // 8)] = 1.0
        out.append(img * (m if k == 0 else torch.clip(m, 0.95, 1.0)))
    return torch.cat(out, 0)


_ORACLE = {}


def _oracle_forward(sd, cfg, B):
    """The oracle's forward of the first B bench patches at t = 417 (cached: bf16 and fp16 share it)."""
    if B not in _ORACLE:
        x = torch.from_numpy(rng.randn((B, 3, H, H), 1, 120))
        cond = _patch_conditions(B)
        tv = torch.full((B,), 417, dtype=torch.long)
        taps = {}
        with torch.no_grad():
            y = unet_ref.unet_forward(sd, cfg, x, cond, tv, taps)
        _ORACLE[B] = (x, cond, tv, y, taps)
    return _ORACLE[B]


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("B", [4, 8])
def test_forward_16bit_at_the_bench_shape_matches_oracle(dtype, B):
    """One denoiser evaluation of B bench patches in 16-bit storage against the fp32 oracle, output and every recorded tap.
    B = 4 is the plan of a timed sub-batch, B = 8 the solo leg's: in both the eight 32->32 @256^2 convolutions run on the lean
    kernel (conv3x3_s32.hip, round 5: with write-through output stores it beats the persistent LDS-DMA kernel at every batch
    size, finding 99, which is retired from the default routing), the four 64->32 ones on the generic kernel (launch counters)."""
    net, sd = _net(dtype)
    x, cond, tv, y_ref, taps = _oracle_forward(sd, net.cfg, B)
    lib = cabi.lib()
    c32, s32 = lib.ld_counter(cabi.COUNTER_CONV3X3_C32), lib.ld_counter(cabi.COUNTER_CONV3X3_S32)
    y = net(x.cuda(), cond.cuda(), tv.cuda()).cpu()
    ran_c32, ran_s32 = lib.ld_counter(cabi.COUNTER_CONV3X3_C32) - c32, lib.ld_counter(cabi.COUNTER_CONV3X3_S32) - s32
    plan = net.plan(B, H, H)
    fams = [(m.get("family", ""), m.get("shape", "")) for m in plan.meta.values()]
    n_c32 = sum(f.startswith("conv3x3_c32") for f, _ in fams)
    n_s32 = sum(f.startswith("conv3x3_s32") for f, _ in fams)
    # (the conditioning encoder's 32->32 convolution at 256^2 runs on the lean kernel too)
    assert ran_c32 == 0 and n_c32 == 0, (ran_c32, n_c32)
    assert ran_s32 >= 8 and n_s32 == 8, (ran_s32, n_s32)
    assert sum(f.startswith("conv3x3_s32") and s == "32->32@256x256" for f, s in fams) == 8, fams
    assert sum(f.startswith("conv3x3<") and s == "64->32@256x256" for f, s in fams) == 4, fams
    worst = ("", 0.0)
    for name, buf in plan.named.items():
        if name not in taps:
            continue
        got = buf.float().permute(0, 3, 1, 2).cpu()
        rel = float((got - taps[name]).abs().max()) / max(float(taps[name].abs().max()), 1e-6)
        if rel > worst[1]:
            worst = (name, rel)
        assert rel < TAP_TOL[dtype], (name, rel)
    rel = float((y - y_ref).abs().max()) / float(y_ref.abs().max())
    mean = float((y - y_ref).abs().mean())
    print(f"3x256^2 B={B} {dtype}: out rel max {rel:.3e}, mean-abs {mean:.3e}; worst tap {worst[0]} {worst[1]:.3e}; "
          f"persistent conv launches {ran_c32}")
    assert torch.isfinite(y).all() and rel < OUT_TOL[dtype], rel


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_chain_16bit_at_the_bench_shape_matches_oracle(dtype):
    """T = 6 reverse steps of the 8 bench patches through the regime bench.py times (two concurrent sub-batches of 4,
    replayed step graphs, device noise) against oracle/diffusion_ref.RefSampler fed the same noise stream: absolute
    mean-abs / max-abs bounds on the final x (image range [0, 2])."""
    B, T = 8, 6
    net, sd = _net(dtype)
    cfg = dict(branch_out=False, start_intermediate=False, start_timestep=2, data="mvtec", mask_x=False, mask_cond=False,
               ood_AD=False, ood_confidence=False, classifier=False, use_gt=False, use_gt_timestep=100)
    gd = ldh.GaussianDiffusion(cfg, net, image_size=H, timesteps=T, beta_schedule="sigmoid", objective="pred_x0",
                               auto_normalize=False).to("cuda")
    gd.noise_source = "device"                     # the counter-based stream of rng.py, drawn on the GPU
    cond = _patch_conditions(B)
    out = gd.sample(cond.cuda(), None, batch_size=B, min_max_val=(0.0, 2.0)).cpu().numpy()
    sub = gd._subs.get((id(net.plan(B, H, H, table_T=T)), 2))
    assert sub is not None and sub.b == 4 and sub.graphs, "the chain did not run as two replayed sub-batches of 4"
    if "chain" not in _ORACLE:                     # bf16 and fp16 share the oracle's run (48 patch-forwards on the host)
        o = diffusion_ref.SamplerOptions(timesteps=T, data="mvtec")
        smp = diffusion_ref.RefSampler(diffusion_ref.make_model_fn(sd, net.cfg), o, 3, H)
        ns = rng.NoiseStream(10)
        with torch.no_grad():
            _ORACLE["chain"] = smp.sample(cond, None, (0.0, 2.0), B, lambda s: torch.from_numpy(ns.next(tuple(s)))).numpy()
    ref = _ORACLE["chain"]
    d = np.abs(out - ref)
    print(f"3x256^2, 8 patches as 2 sub-batches of 4, T={T}, {dtype}: mean-abs {d.mean():.3e} max-abs {d.max():.3e} vs oracle")
    assert np.isfinite(out).all() and out.min() >= 0.0 and out.max() <= 2.0
    assert d.mean() <= CHAIN_MEAN[dtype] and d.max() <= CHAIN_MAX[dtype], (float(d.mean()), float(d.max()))
    # run to run: the only order-dependent sums of the path are the GroupNorm statistics (fp64 atomics over 16 stripes), whose
    # 1e-16 differences do not reach an fp32 coefficient: the same call again gives the same image, bit for bit
    again = gd.sample(cond.cuda(), None, batch_size=B, min_max_val=(0.0, 2.0)).cpu().numpy()
    assert np.array_equal(out, again), float(np.abs(out - again).max())
