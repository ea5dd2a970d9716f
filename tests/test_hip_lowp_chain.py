"""Chain-level parity of the 16-bit storage modes (bf16, fp16) -- the modes bench.py runs -- against golden vectors
generated from the real reference (fp32 CPU): the full T=1000 DDPM chain of BASELINE.json configs[1] (G5, one 128x128
patch) and the S=50-of-1000 DDIM branch + fusion run (G7), with the fixtures' noise stream.

The north star's 1e-3 max-abs gate is stated for fp32 and met there (tests/test_hip_sampler.py).  16-bit storage
cannot meet it, so each mode has its OWN stated bounds (also quoted in DESIGN.md section 2), in units of the image
range [0, 2]:

  * up to t = 100 (900 of the 1000 steps) the chain is well conditioned and the bounds are tight;
  * over the last 100 steps this random-init network amplifies ANY perturbation by two orders of magnitude -- the
    fp32 HIP path itself goes from 1.5e-6 (t=100) to 4.4e-4 (t=0) against the reference, and the reference's own fp32
    code with nothing but its denoiser OUTPUT rounded to 16 bits once per step (fixture G11, made by
    tools/make_goldens.py from the real reference) goes from 5e-5 to 3.8e-3 mean-abs (bf16).  The final-state bounds
    are therefore stated twice: absolutely, and as "the HIP path's error growth over the last 100 steps is no worse
    than 2x the growth of that reference-derived run", which separates kernel accuracy from chain conditioning."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from localdiffusion_hallucination_amd import rng                  # noqa: E402
from test_hip_sampler import make, run                            # noqa: E402

RECORDS = (999, 750, 500, 250, 100, 10, 0)
# (max-abs, mean-abs); measured on MI355X (round 2): bf16 t=100 2.9e-2 / 3.2e-3, final 1.98 / 0.128;
#                                                    fp16 t=100 3.9e-3 / 4.6e-4, final 0.95 / 2.0e-2
# Round 3 (tools/exp_error_budget.py, profiles/r03_error_budget_*.txt): with the roundings emulated on the oracle one
# storage class at a time, rounding the WEIGHTS alone gives 3.39e-2 / 4.38e-3 (bf16) and 3.51e-3 / 3.93e-4 (fp16) at
# t = 100 -- the whole of the "all classes" figure (3.39e-2 / 4.41e-3, 3.83e-3 / 3.98e-4); every activation class is
# 10-20x below (all of them together 6.4e-3 / 5.0e-4 in bf16).  The distance to the reference is the distance between
# the network with fp32 weights and the network with 16-bit weights, not error the kernels accumulate: the bound is
# therefore stated against that figure (W_ONLY_T100 x 1.5) and tightened from round 2's 6e-2 / 6e-3, 8e-3 / 1e-3.
W_ONLY_T100 = {"bf16": (3.39e-2, 4.38e-3), "fp16": (3.51e-3, 3.93e-4)}
BOUND_T100 = {"bf16": (4.5e-2, 4.5e-3), "fp16": (5.3e-3, 5.9e-4)}
BOUND_FINAL_MEAN = {"bf16": 0.2, "fp16": 4e-2}
# DDIM S=50 of 1000 (G7, 64x64, branch + fusion at times[-4]): measured bf16 0.33 / 1.8e-2, fp16 0.54 / 8.3e-3
BOUND_G7_MEAN = {"bf16": 4e-2, "fp16": 2e-2}


def err(got, ref):
    d = np.abs(got - ref)
    return float(d.max()), float(d.mean())


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_cfg2_chain_16bit_vs_reference_golden(golden, dtype):
    g, floor = golden("g5_cfg2_mri128"), golden("g11_cfg2_output_rounded")
    cond = torch.from_numpy(rng.uniform((1, 1, 128, 128), 5, 1, 0.0, 2.0))
    gd = make(dict(mode="mri"), dict(data="mri"), 128, 1000, dtype=dtype)
    hist = gd.sample(cond.cuda(), None, batch_size=1, min_max_val=(0.0, 2.0), return_all_timesteps=True).cpu().numpy()
    assert hist.shape == (1, 1001, 1, 128, 128) and np.isfinite(hist).all()
    e, f = {}, {}
    for t in RECORDS:
        ref = g[f"x_after_t{t}"]
        e[t] = err(hist[:, 1000 - t], ref)
        f[t] = err(floor[f"{dtype}_x_after_t{t}"], ref)
        print(f"G5 cfg2 {dtype} x after t={t:3d}: HIP max-abs {e[t][0]:.3e} mean-abs {e[t][1]:.3e} | reference with its "
              f"output rounded to {dtype}: max-abs {f[t][0]:.3e} mean-abs {f[t][1]:.3e}")
    for t in (999, 750, 500, 250, 100):
        assert e[t][0] <= BOUND_T100[dtype][0] and e[t][1] <= BOUND_T100[dtype][1], (dtype, t, e[t])
    # the kernels add (almost) nothing to what rounding the weights to 16 bits costs
    assert e[100][1] <= 1.5 * W_ONLY_T100[dtype][1], (dtype, e[100], W_ONLY_T100[dtype])
    assert e[0][1] <= BOUND_FINAL_MEAN[dtype], (dtype, e[0])
    growth_hip, growth_ref = e[0][1] / e[100][1], f[0][1] / f[100][1]
    print(f"G5 cfg2 {dtype}: mean-abs growth over the last 100 steps: HIP x{growth_hip:.0f}, output-rounded reference x{growth_ref:.0f}")
    assert growth_hip <= 2.0 * growth_ref, (dtype, growth_hip, growth_ref)


# Two-term weights on the full- and half-resolution layers -- 3x3 / 1x1 convolutions and the linear-attention projections
# (Unet.set_weight_split_levels(2), +10 % step time): measured round 3 at t = 100: bf16 7.4e-3 / 5.6e-4 (one-term
# 2.6e-2 / 3.2e-3), fp16 7.1e-4 / 7.0e-5 (4.0e-3 / 4.6e-4); what is left is the activation roundings' floor
BOUND_T100_SPLIT = {"bf16": (1.4e-2, 9e-4), "fp16": (1.5e-3, 1.2e-4)}


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_cfg2_chain_with_two_term_weights(golden, dtype):
    """The accuracy mode of the 16-bit storage types: W = hi + lo for the convolutions of the first two resolution levels
    (the weights whose rounding tools/exp_error_budget.py found to carry 90 % of the chain's distance to the reference).
    Same run and golden as above; the distance must come down by at least 3.5x at t = 100 and stay below its own bounds."""
    g = golden("g5_cfg2_mri128")
    cond = torch.from_numpy(rng.uniform((1, 1, 128, 128), 5, 1, 0.0, 2.0))
    gd = make(dict(mode="mri"), dict(data="mri"), 128, 1000, dtype=dtype)
    gd.model.set_weight_split_levels(2)
    hist = gd.sample(cond.cuda(), None, batch_size=1, min_max_val=(0.0, 2.0), return_all_timesteps=True).cpu().numpy()
    assert np.isfinite(hist).all()
    e = {t: err(hist[:, 1000 - t], g[f"x_after_t{t}"]) for t in RECORDS}
    for t in RECORDS:
        print(f"G5 cfg2 {dtype}, two-term weights on 2 levels, x after t={t:3d}: max-abs {e[t][0]:.3e} mean-abs {e[t][1]:.3e}")
    for t in (999, 750, 500, 250, 100):
        assert e[t][0] <= BOUND_T100_SPLIT[dtype][0] and e[t][1] <= BOUND_T100_SPLIT[dtype][1], (dtype, t, e[t])
    assert e[100][1] <= W_ONLY_T100[dtype][1] / 3.5, (dtype, e[100])
    assert e[0][1] <= 0.5 * BOUND_FINAL_MEAN[dtype], (dtype, e[0])


# ---- the end-to-end pin of the storage modes on a WELL-CONDITIONED denoiser (golden G16) ----------------
# cfg2's shape and schedule run by the real reference on the contractive procedural net (final_conv gain 0.25: a perturbation
# grows x1.7 in the mean over the last 100 steps and not at all over the last 10, tools/exp_contractive.py), so the distance
# of the FINAL image is a statement about the implementation and is bounded in MAX-abs.  Bounds (max-abs, mean-abs) on the
# [0, 2] range through t = 100 and at t <= 10.  Round 6: the 16-bit bounds sit at 1.3x what MI355X measures
# (they were ~2x: a summation-order change that doubled the distance would have passed silently; finding 105 had moved the bf16
# final image from 9.3e-3 to 1.1e-2 inside the old bound) -- the next such change is a conscious decision.  The values repeat to
# every printed digit from run to run and box to box (fp64 statistics, fixed summation orders); fp32 keeps 2x because its distance
# is a few ulps of the image (quantised in steps of 1.2e-7).  Measured (printed by the test):
#             through t = 100          final image (t = 0)
#   fp32      1.2e-6 / 8.6e-8          2.2e-6 / 2.0e-7     (the reference against itself, 1 thread vs 8: 1.4e-6 / 1.9e-7)
#   bf16      2.1e-3 / 3.8e-4          1.17e-2 / 7.5e-4    (before the side res_conv, finding 117: 2.3e-3 / 3.7e-4 and 1.125e-2 / 7.4e-4)
#   fp16      3.9e-4 / 5.6e-5          2.0e-3 / 1.1e-4
#   bf16x2    4.6e-4 / 5.1e-5          9.8e-3 / 6.9e-4     (two-term weights on two levels)
#   fp16x2    5.2e-5 / 6.6e-6          1.4e-3 / 8.7e-5
# What the last ten steps add in every 16-bit mode is the rounding of ONE evaluation's activations: the posterior weight of
# x0_hat goes to 1 as t -> 0, so the final image carries the denoiser output's own 16-bit error (a single forward at the
# bench shape: 6.7e-3 of the range in bf16, 9e-4 in fp16, tests/test_hip_bench_shape.py) -- two-term WEIGHTS remove the
# chain's accumulated part (t = 100: 5x / 7x closer) and leave that one untouched.
G16_BOUNDS = {"fp32": ((2.4e-6, 1.8e-7), (4.5e-6, 4e-7)), "bf16": ((2.95e-3, 4.9e-4), (1.5e-2, 9.6e-4)), "fp16": ((4.9e-4, 7.3e-5), (2.7e-3, 1.4e-4)),
              "bf16x2": ((6.0e-4, 6.7e-5), (1.3e-2, 9.0e-4)), "fp16x2": ((6.8e-5, 8.5e-6), (1.76e-3, 1.13e-4))}


@pytest.mark.parametrize("mode", ["fp32", "bf16", "fp16", "bf16x2", "fp16x2"])
def test_cfg2_contractive_chain_pins_every_storage_mode(golden, mode):
    g = golden("g16_cfg2_contractive")
    dtype = mode[:4]
    cond = torch.from_numpy(rng.uniform((1, 1, 128, 128), 5, 1, 0.0, 2.0))
    gd = make(dict(mode="mri"), dict(data="mri"), 128, 1000, dtype=dtype, final_gain=float(g["final_gain"]))
    if mode.endswith("x2"):
        gd.model.set_weight_split_levels(2)
    hist = gd.sample(cond.cuda(), None, batch_size=1, min_max_val=(0.0, 2.0), return_all_timesteps=True).cpu().numpy()
    assert hist.shape == (1, 1001, 1, 128, 128) and np.isfinite(hist).all()
    e = {t: err(hist[:, 1000 - t], g[f"x_after_t{t}"]) for t in RECORDS}
    for t in RECORDS:
        print(f"G16 contractive cfg2, {mode:6s} x after t={t:3d}: max-abs {e[t][0]:.3e} mean-abs {e[t][1]:.3e}   "
              f"(reference, 1 thread vs all: {float(g[f'self_maxabs_t{t}']):.3e} / {float(g[f'self_meanabs_t{t}']):.3e})")
    (b100, b0) = G16_BOUNDS[mode]
    for t in (999, 750, 500, 250, 100):
        assert e[t][0] <= b100[0] and e[t][1] <= b100[1], (mode, t, e[t])
    for t in (10, 0):
        assert e[t][0] <= b0[0] and e[t][1] <= b0[1], (mode, t, e[t])
    assert err(hist[:, -1], g["final"]) == e[0]
    if not mode.endswith("x2"):
        # the chain is contractive: what the last 100 steps add stays within a small factor of what was there at t = 100
        # (gain 3, G5 / G11: x40-70).  With two-term weights the t = 100 distance is so small that the final evaluation's
        # activation rounding dominates the end: bounded absolutely above.
        assert e[0][1] <= 3.0 * max(e[100][1], 1e-7), (mode, e[100], e[0])


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_g7_ddim_chain_16bit_vs_reference_golden(golden, dtype):
    g = golden("g7_ddim")
    cond = torch.from_numpy(rng.uniform((1, 1, 64, 64), 7, 1, 0.0, 2.0))
    mask = torch.from_numpy(g["mask"])
    kw = dict(data="mri", branch_out=True, start_intermediate=True, start_timestep=2, mask_x=True)
    mx, mean = err(run(make(dict(mode="mri"), kw, 64, 1000, 50, dtype=dtype), cond, mask, 1), g["fused_final"])
    print(f"G7 DDIM S=50 of 1000, branch + fusion, {dtype}: max-abs {mx:.3e} mean-abs {mean:.3e}")
    assert mean <= BOUND_G7_MEAN[dtype], (dtype, mx, mean)
    kw = dict(data="mri", branch_out=True, start_intermediate=False, mask_x=True)
    mx, mean = err(run(make(dict(mode="mri"), kw, 64, 50, 10, dtype=dtype), cond, mask, 1), g["nofuse_final"])
    print(f"G7 DDIM S=10 of 50, branches kept apart, {dtype}: max-abs {mx:.3e} mean-abs {mean:.3e}")
    assert mean <= BOUND_G7_MEAN[dtype], (dtype, mx, mean)


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_cfg1_and_branch_fusion_16bit_vs_reference_golden(golden, dtype):
    """BASELINE.json configs[0] (MNIST 28x28, T=100, 4 patches, branch + fusion at t <= 2; G4) and the T=50 branch +
    fusion runs of G6 (mri: OOD prediction kept; mnist: replaced by the conditioning) in 16-bit storage against the
    reference goldens.  Printed per case; asserted on the mean (measured: see DESIGN.md section 2)."""
    from test_hip_sampler import MNIST
    bound = {"bf16": 6e-2, "fp16": 1.5e-2}[dtype]
    g = golden("g4_cfg1_mnist")
    gd = make(MNIST, dict(data="mnist", branch_out=True, start_intermediate=True, start_timestep=2, mask_x=True, ood_AD=True),
              28, 100, dtype=dtype)
    mx, mean = err(run(gd, torch.from_numpy(g["cond"]), torch.from_numpy(g["mask"]), 4), g["final"])
    print(f"G4 cfg1 (MNIST, T=100, 4 patches, branch + fusion) {dtype}: max-abs {mx:.3e} mean-abs {mean:.3e}")
    assert mean <= bound, (dtype, mx, mean)
    g6 = golden("g6_branch_fusion")
    for tag, kw, H, data in [("mri32", dict(mode="mri"), 32, "mri"), ("mnist28", MNIST, 28, "mnist")]:
        cond = torch.from_numpy(rng.uniform((2, 1, H, H), 6, 1, 0.0, 2.0))
        mask = torch.zeros(2, 1, H, H)
        mask[:, :, :, :H // 4] = 1.0
        gd = make(kw, dict(data=data, branch_out=True, start_intermediate=True, start_timestep=2, mask_x=True), H, 50, dtype=dtype)
        mx, mean = err(run(gd, cond, mask, 2), g6[tag + "_final"])
        print(f"G6 {tag} (T=50, branch + fusion) {dtype}: max-abs {mx:.3e} mean-abs {mean:.3e}")
        assert mean <= bound, (dtype, tag, mx, mean)


def test_two_term_weights_at_the_bench_shape():
    """The accuracy mode at BASELINE.json configs[2]'s shape (3x256x256 patches, the 12.1M-parameter denoiser), where no
    reference golden exists: a T = 300 chain of two patches in fp32 storage on the HIP path (the mode the 1e-3 gate pins to
    the oracle) against the same chain in bf16 with one-term and with two-term weights on two levels.  Size-independent
    property: the two-term chain is the closer one at every recorded state before the chaotic tail."""
    from localdiffusion_hallucination_amd import weights
    import localdiffusion_hallucination_amd as ldh
    H, B, T = 256, 2, 300
    cond = torch.from_numpy(rng.uniform((B, 3, H, H), 33, 1, 0.0, 2.0))
    cfg = dict(branch_out=False, start_intermediate=False, start_timestep=2, data="mvtec", mask_x=False, mask_cond=False,
               ood_AD=False, ood_confidence=False, classifier=False, use_gt=False, use_gt_timestep=100)
    hist = {}
    for tag, dtype, levels in (("fp32", "fp32", 0), ("bf16", "bf16", 0), ("bf16x2", "bf16", 2)):
        net = ldh.Unet(dim=32, init_dim=32, channels=3, out_dim=3, mode="mvtec", compute_dtype=dtype)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in weights.procedural_state_dict(net.cfg, 0).items()})
        net.set_weight_split_levels(levels)
        gd = ldh.GaussianDiffusion(cfg, net, image_size=H, timesteps=T, beta_schedule="sigmoid", objective="pred_x0",
                                   auto_normalize=False).to("cuda")
        gd.noise_source = "device"
        hist[tag] = gd.sample(cond.cuda(), None, batch_size=B, min_max_val=(0.0, 2.0), return_all_timesteps=True).cpu().numpy()
        assert np.isfinite(hist[tag]).all()
    for t in (250, 150, 100, 50):
        e1 = float(np.abs(hist["bf16"][:, T - t] - hist["fp32"][:, T - t]).mean())
        e2 = float(np.abs(hist["bf16x2"][:, T - t] - hist["fp32"][:, T - t]).mean())
        print(f"3x256^2, T={T}, x after t={t:3d}: bf16 one-term {e1:.3e}, two-term weights on 2 levels {e2:.3e} (mean-abs vs fp32 storage)")
        assert e2 < 0.45 * e1, (t, e1, e2)          # measured (round 3): 0.33, 0.24, 0.23, 0.23
