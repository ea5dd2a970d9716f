"""The oracle against the committed golden vectors (made by tools/make_goldens.py from the real
reference).  CPU only; sized to finish in a couple of minutes."""
import numpy as np
import pytest
import torch

from localdiffusion_hallucination_amd import rng, schedule, weights
from oracle import diffusion_ref, unet_ref

CFG_MNIST = weights.UnetConfig(dim_mults=(1, 2, 4), full_attn=(False, False, True), mode="mnist")
CFG_MRI = weights.UnetConfig(mode="mri")
CFG_MVTEC = weights.UnetConfig(channels=3, out_dim=3, mode="mvtec")


def sd_of(cfg):
    return {k: torch.from_numpy(v) for k, v in weights.procedural_state_dict(cfg, 0).items()}


class Noise:
    def __init__(self, seed=10):
        self.s = rng.NoiseStream(seed)

    def __call__(self, shape):
        return torch.from_numpy(self.s.next(tuple(shape)))


def test_inventory_counts():
    txt = open(__import__("os").path.join(__import__("conftest").GOLDEN, "g0_inventory.txt")).read().split("\n")
    want = {l.split()[0]: (int(l.split()[1]), int(l.split()[2])) for l in txt if l}
    for tag, cfg in [("mnist", CFG_MNIST), ("mri", CFG_MRI), ("mvtec", CFG_MVTEC)]:
        sh = weights.unet_param_shapes(cfg)
        assert (len(sh), weights.num_params(cfg)) == want[tag]


@pytest.mark.parametrize("sched", ["sigmoid", "linear", "cosine"])
@pytest.mark.parametrize("T", [50, 100, 1000])
def test_schedule_buffers(golden, sched, T):
    g = golden("g1_schedules")
    ours = schedule.make_buffers(T, sched, "pred_x0")
    orc = diffusion_ref.schedule_buffers(sched, T, "pred_x0")
    for name in schedule.BUFFER_NAMES:
        ref = g[f"{sched}_{T}_{name}"]
        assert np.array_equal(ours[name].numpy(), ref), name           # product: bit-exact
        np.testing.assert_allclose(orc[name].numpy(), ref, rtol=3e-7, atol=1e-37)


def test_schedule_known_answers():
    """SURVEY.md 8a-11 KATs (sigmoid T=1000 / T=100, linear T=1000)."""
    b = schedule.make_buffers(1000, "sigmoid")
    assert abs(float(b["betas"][0]) - 3.0027919741e-4) < 1e-10
    assert abs(float(b["betas"][-1]) - 0.999) < 1e-7
    assert abs(float(b["alphas_cumprod"][-1]) - 3.0028698680e-7) < 1e-12
    assert abs(float(b["posterior_log_variance_clipped"][0]) - (-46.051702)) < 1e-4
    assert abs(float(b["posterior_log_variance_clipped"][1]) - (-8.800935)) < 1e-4
    assert float(b["posterior_mean_coef1"][0]) == 1.0 and float(b["posterior_mean_coef2"][0]) == 0.0
    assert abs(float(b["posterior_mean_coef1"][-1]) - 1.73114671e-2) < 1e-8
    b = schedule.make_buffers(100, "sigmoid")
    assert abs(float(b["betas"][0]) - 3.0772859361e-3) < 1e-9
    b = schedule.make_buffers(1000, "linear")
    assert abs(float(b["betas"][0]) - 1e-4) < 1e-10 and abs(float(b["betas"][-1]) - 2e-2) < 1e-8
    assert abs(float(b["alphas_cumprod"][-1]) - 4.0358297654e-5) < 1e-10


@pytest.mark.parametrize("tag,cfg,ts", [("mnist28", CFG_MNIST, (0, 5, 99)),
                                        ("mri64", CFG_MRI, (0, 999)),
                                        ("mvtec32", CFG_MVTEC, (3, 777))])
def test_unet_forward(golden, tag, cfg, ts):
    g = golden("g2_unet_forward")
    B, C, H, cin = [int(v) for v in g[f"{tag}_shape"]]
    sd = sd_of(cfg)
    x = torch.from_numpy(rng.randn((B, C, H, H), 1, 100))
    cond = torch.from_numpy(rng.uniform((B, cin, H, H), 1, 101, 0.0, 2.0))
    for t in ts:
        taps = {}
        with torch.no_grad():
            y = unet_ref.unet_forward(sd, cfg, x, cond, torch.full((B,), t, dtype=torch.long), taps)
        np.testing.assert_allclose(y.numpy(), g[f"{tag}_t{t}_out"], atol=2e-5, rtol=0)
        for key in g.files:
            pre = f"{tag}_t{t}_tap_"
            if key.startswith(pre):
                tt = taps[key[len(pre):]].float()
                assert abs(float(tt.mean()) - g[key][0]) < 1e-4
                assert abs(float(tt.norm()) - g[key][1]) < 1e-3 * max(1.0, g[key][1])


def _run(cfg, H, B, T, S, cond, mask, **kw):
    o = diffusion_ref.SamplerOptions(timesteps=T, sampling_timesteps=S, **kw)
    smp = diffusion_ref.RefSampler(diffusion_ref.make_model_fn(sd_of(cfg), cfg), o, cfg.channels, H)
    with torch.no_grad():
        out = smp.sample(cond, mask, (0.0, 2.0), B, Noise(10))
    return np.stack([t.numpy() for t in out]) if isinstance(out, list) else out.numpy()


def test_three_step_runs(golden):
    g = golden("g3_three_step")
    cond, mask = torch.from_numpy(g["cond"]), torch.from_numpy(g["mask"])
    cases = {
        "nonbranch": dict(data="mnist"),
        "branch_fuse_mnist": dict(data="mnist", branch_out=True, start_intermediate=True, start_timestep=1, mask_x=True),
        "branch_fuse_mri": dict(data="mri", branch_out=True, start_intermediate=True, start_timestep=1, mask_x=True),
        "branch_nofuse": dict(data="mri", branch_out=True, start_intermediate=False, mask_x=True),
    }
    for tag, kw in cases.items():
        out = _run(CFG_MNIST, 28, 2, 3, None, cond, mask, **kw)
        assert out.shape == g[tag].shape
        np.testing.assert_allclose(out, g[tag], atol=1e-5, rtol=0)


def test_cfg1_mnist_full(golden):
    """BASELINE.json configs[0]: MNIST 28x28, T=100, 4 patches, branch + fusion."""
    g = golden("g4_cfg1_mnist")
    out = _run(CFG_MNIST, 28, 4, 100, None, torch.from_numpy(g["cond"]), torch.from_numpy(g["mask"]),
               data="mnist", branch_out=True, start_intermediate=True, start_timestep=2, mask_x=True, ood_AD=True)
    np.testing.assert_allclose(out, g["final"], atol=5e-5, rtol=0)


def test_ddim(golden):
    g = golden("g7_ddim")
    cond = torch.from_numpy(rng.uniform((1, 1, 64, 64), 7, 1, 0.0, 2.0))
    mask = torch.from_numpy(g["mask"])
    out = _run(CFG_MRI, 64, 1, 50, 10, cond, mask, data="mri", branch_out=True, start_intermediate=False, mask_x=True)
    np.testing.assert_allclose(out, g["nofuse_final"], atol=5e-5, rtol=0)
    out = _run(CFG_MRI, 64, 1, 50, 10, cond, None, data="mri")
    np.testing.assert_allclose(out, g["single_final"], atol=5e-5, rtol=0)


def test_fallback_and_objectives(golden):
    g = golden("g8_fallback_objectives")
    cond = torch.from_numpy(rng.uniform((2, 1, 28, 28), 8, 1, 0.0, 2.0))
    out = _run(CFG_MNIST, 28, 2, 20, None, cond, torch.ones(2, 1, 28, 28), data="mnist", branch_out=True,
               start_intermediate=True, start_timestep=2, mask_x=True)
    np.testing.assert_allclose(out, g["allones_final"], atol=1e-5, rtol=0)
    for obj in ("pred_noise", "pred_v"):
        out = _run(CFG_MNIST, 28, 2, 20, None, cond, None, data="mnist", objective=obj)
        np.testing.assert_allclose(out, g[obj + "_final"], atol=1e-4, rtol=0)


class StubClassifier:
    """Same stand-in as tools/make_goldens.py: score -1 for the first ``reject`` calls, +1 afterwards."""
    def __init__(self, reject):
        self.reject, self.calls = reject, 0

    def __call__(self, x0):
        self.calls += 1
        return (torch.tensor(-1.0 if self.calls <= self.reject else 1.0), None, None)


@pytest.mark.parametrize("tag,cfg,H,data,key", [("mnist28_reject2", CFG_MNIST, 28, "mnist", "28"),
                                                ("mri32_reject3", CFG_MRI, 32, "mri", "32"),
                                                ("mri32_reject_all", CFG_MRI, 32, "mri", "32")])
def test_classifier_gated_rebranching(golden, tag, cfg, H, data, key):
    """fusion() (ddpm.py:883-916): rejected joint steps are redone as branch + fusion steps from the masked
    branch states; t == 0 always accepts.  Golden = the real reference with the same stub classifier."""
    g = golden("g9_classifier_gate")
    cond, mask = torch.from_numpy(g["cond" + key]), torch.from_numpy(g["mask" + key])
    calls, reject = (int(v) for v in g[tag + "_calls"])
    o = diffusion_ref.SamplerOptions(timesteps=12, branch_out=True, start_intermediate=True, start_timestep=7,
                                     data=data, mask_x=True, classifier=True)
    smp = diffusion_ref.RefSampler(diffusion_ref.make_model_fn(sd_of(cfg), cfg), o, cfg.channels, H)
    smp.classifier = StubClassifier(reject)
    with torch.no_grad():
        out = smp.sample(cond, mask, (0.0, 2.0), 2, Noise())
    assert smp.classifier.calls == calls
    assert float(np.abs(out.numpy() - g[tag + "_final"]).max()) <= 1e-6


def _x0_arrays(lst):
    pairs = [np.stack([e[0].numpy(), e[1].numpy()]) for e in lst if isinstance(e, (list, tuple))]
    singles = [e.numpy() for e in lst if not isinstance(e, (list, tuple))]
    return pairs, singles


def test_use_gt_start_and_return_all(golden):
    """use_gt start (ddpm.py:937-944) + return_all_timesteps / return_all_outputs (:946, 959-977, 1072):
    golden = the real reference called with those flags."""
    g = golden("g10_use_gt_return_all")

    def smp(cfg, H, T, S=None, **kw):
        o = diffusion_ref.SamplerOptions(timesteps=T, sampling_timesteps=S, **kw)
        return diffusion_ref.RefSampler(diffusion_ref.make_model_fn(sd_of(cfg), cfg), o, cfg.channels, H)
    with torch.no_grad():
        ret, x0s, cm = smp(CFG_MNIST, 28, 50, data="mnist", start_intermediate=True, use_gt=True, use_gt_timestep=20).sample(
            torch.from_numpy(g["a_cond"]), None, (0.0, 2.0), 2, Noise(), gt=torch.from_numpy(g["a_hr"]),
            return_all_timesteps=True, return_all_outputs=True)
    assert cm == [] and len(x0s) == 20
    np.testing.assert_allclose(ret.numpy(), g["a_hist"], atol=1e-6, rtol=0)
    np.testing.assert_allclose(np.stack([e.numpy() for e in x0s]), g["a_x0"], atol=1e-6, rtol=0)
    with torch.no_grad():
        ret, x0s, _ = smp(CFG_MRI, 32, 40, data="mri", branch_out=True, start_intermediate=True, start_timestep=3, mask_x=True,
                          use_gt=True, use_gt_timestep=12).sample(
            torch.from_numpy(g["b_cond"]), torch.from_numpy(g["b_mask"]), (0.0, 2.0), 2, Noise(),
            gt=torch.from_numpy(g["b_hr"]), return_all_outputs=True)
    pairs, singles = _x0_arrays(x0s)
    np.testing.assert_allclose(ret.numpy(), g["b_final"], atol=1e-6, rtol=0)
    np.testing.assert_allclose(np.stack(pairs), g["b_x0_pairs"], atol=1e-6, rtol=0)
    np.testing.assert_allclose(np.stack(singles), g["b_x0_single"], atol=1e-6, rtol=0)
    with torch.no_grad():
        ret = smp(CFG_MNIST, 28, 50, 10, data="mnist").sample(torch.from_numpy(g["c_cond"]), None, (0.0, 2.0), 2, Noise(),
                                                              return_all_timesteps=True)
    np.testing.assert_allclose(ret.numpy(), g["c_hist"], atol=1e-6, rtol=0)
    with pytest.raises(TypeError):         # per-branch [out, in] lists cannot be stacked: the reference raises here too
        smp(CFG_MRI, 32, 40, data="mri", branch_out=True, start_intermediate=True, start_timestep=3, mask_x=True).sample(
            torch.from_numpy(g["b_cond"]), torch.from_numpy(g["b_mask"]), (0.0, 2.0), 2, Noise(), return_all_timesteps=True)


def test_kmask_generalisation_reduces_to_the_reference_for_two_branches(golden):
    """the K-mask branch -> fusion loop of the oracle, run with K = 2 and masks [m, 1 - (m >= 1)], must
    give the reference's two-branch goldens BIT FOR BIT (G6: fusion at t <= 2, mri = OOD prediction kept, mnist = OOD
    prediction replaced by cond_out; G3: branches kept apart)."""
    g = golden("g6_branch_fusion")
    for tag, cfg, H, data in [("mri32", CFG_MRI, 32, "mri"), ("mnist28", CFG_MNIST, 28, "mnist")]:
        cond = torch.from_numpy(rng.uniform((2, 1, H, H), 6, 1, 0.0, 2.0))
        mask = torch.zeros(2, 1, H, H)
        mask[:, :, :, :H // 4] = 1.0
        masks = torch.cat([mask, 1.0 - (mask >= 1.0).float()], 1)
        o = diffusion_ref.SamplerOptions(timesteps=50, branch_out=True, start_intermediate=True, start_timestep=2, data=data, mask_x=True)
        fresh = lambda: diffusion_ref.RefSampler(diffusion_ref.make_model_fn(sd_of(cfg), cfg), o, 1, H)   # mask_x carries over
        with torch.no_grad():
            out = fresh().p_sample_loop_kmask(cond, masks, (0.0, 2.0), (2, 1, H, H), Noise(), True, True)
            two = fresh().sample(cond, mask, (0.0, 2.0), 2, Noise())
        assert torch.equal(out, two), tag                                  # same operations: bit-equal to the 2-branch oracle
        np.testing.assert_allclose(out.numpy(), g[tag + "_final"], atol=1e-6, rtol=0)    # and that is the reference golden
    g3 = golden("g3_three_step")
    cond, mask = torch.from_numpy(g3["cond"]), torch.from_numpy(g3["mask"])
    masks = torch.cat([mask, 1.0 - (mask >= 1.0).float()], 1)
    o = diffusion_ref.SamplerOptions(timesteps=3, branch_out=True, start_intermediate=False, data="mri", mask_x=True)
    smp = diffusion_ref.RefSampler(diffusion_ref.make_model_fn(sd_of(CFG_MNIST), CFG_MNIST), o, 1, 28)
    with torch.no_grad():
        out = smp.p_sample_loop_kmask(cond, masks, (0.0, 2.0), (2, 1, 28, 28), Noise(), False, True)
    np.testing.assert_allclose(out.numpy(), g3["branch_nofuse"], atol=1e-6, rtol=0)


def test_kmask_ddim_histories_and_gate_reduce_to_the_reference_for_two_branches(golden):
    """Round 3: the K-mask forms of the DDIM loop, of return_all_outputs and of the classifier gate, run with K = 2 and
    masks [m, 1 - (m >= 1)], are the reference's two-branch paths bit for bit: G7 (DDIM 50 of 1000, fusion; DDIM 10
    of 50 kept apart), G10 (history of a branch + fusion run), G9 (gate with a rejecting stub classifier)."""
    g = golden("g7_ddim")
    mask = torch.from_numpy(g["mask"])
    H = mask.shape[-1]
    cond = torch.from_numpy(rng.uniform((1, 1, H, H), 7, 1, 0.0, 2.0))
    masks = torch.cat([mask, 1.0 - (mask >= 1.0).float()], 1)
    mk = lambda **kw: diffusion_ref.RefSampler(diffusion_ref.make_model_fn(sd_of(CFG_MRI), CFG_MRI),
                                               diffusion_ref.SamplerOptions(data="mri", branch_out=True, mask_x=True, **kw), 1, H)
    with torch.no_grad():
        out = mk(timesteps=1000, sampling_timesteps=50, start_intermediate=True, start_timestep=2).ddim_sample_kmask(
            cond, masks, (0.0, 2.0), (1, 1, H, H), Noise(), True, True)
        np.testing.assert_allclose(out.numpy(), g["fused_final"], atol=1e-6, rtol=0)
        out = mk(timesteps=50, sampling_timesteps=10, start_intermediate=False).ddim_sample_kmask(
            cond, masks, (0.0, 2.0), (1, 1, H, H), Noise(), False, True)
        np.testing.assert_allclose(np.stack([t.numpy() for t in out]), g["nofuse_final"], atol=1e-6, rtol=0)
    # classifier gate (G9) and the x0 history: K = 2 == the two-branch oracle, which is pinned to the reference
    g9 = golden("g9_classifier_gate")
    cond, mask = torch.from_numpy(g9["cond32"]), torch.from_numpy(g9["mask32"])
    masks = torch.cat([mask, 1.0 - (mask >= 1.0).float()], 1)

    class Stub:
        def __init__(self, reject):
            self.reject, self.calls = reject, 0

        def __call__(self, x0):
            self.calls += 1
            return (torch.tensor(-1.0 if self.calls <= self.reject else 1.0), None, None)
    o = diffusion_ref.SamplerOptions(timesteps=12, branch_out=True, start_intermediate=True, start_timestep=7, data="mri",
                                     mask_x=True, classifier=True)
    smp = diffusion_ref.RefSampler(diffusion_ref.make_model_fn(sd_of(CFG_MRI), CFG_MRI), o, 1, 32)
    smp.classifier = Stub(3)
    with torch.no_grad():
        ret, x0s, _ = smp.p_sample_loop_kmask(cond, masks, (0.0, 2.0), (2, 1, 32, 32), Noise(), True, True, return_all_outputs=True)
    np.testing.assert_allclose(ret.numpy(), g9["mri32_reject3_final"], atol=1e-6, rtol=0)
    assert smp.classifier.calls == int(g9["mri32_reject3_calls"][0])
    smp2 = diffusion_ref.RefSampler(diffusion_ref.make_model_fn(sd_of(CFG_MRI), CFG_MRI), o, 1, 32)
    smp2.classifier = Stub(3)
    with torch.no_grad():
        ret2, x0s2, _ = smp2.sample(cond, mask, (0.0, 2.0), 2, Noise(), return_all_outputs=True)
    assert torch.equal(ret, ret2) and len(x0s) == len(x0s2) == 12
    for a, b in zip(x0s, x0s2):
        if isinstance(b, list):
            assert isinstance(a, list) and all(torch.equal(u, v) for u, v in zip(a, b))
        else:
            assert torch.equal(a, b)


def test_reference_self_distance_fixture(golden):
    """G13: the real reference on ONE thread vs the all-cores golden G5 (same code, weights, noise): the reproducibility
    floor of the 1e-3 gate on cfg2.  The numbers are data (made in the build container); this checks they are what
    DESIGN.md quotes: ~1e-6 after 900 steps, a few 1e-4 at the end -- the reference has ~3x headroom against itself."""
    d = golden("g13_cfg2_reference_self_distance")
    g5 = golden("g5_cfg2_mri128")
    assert list(d["threads"]) [0] == 1
    assert 5e-7 < float(d["maxabs_t100"]) < 5e-6
    assert 1e-4 < float(d["maxabs_t0"]) < 1e-3
    assert abs(float(np.abs(d["final_1thread"] - g5["final"]).max()) - float(d["maxabs_t0"])) < 1e-9


G14_CASES = [("ddpm_maskx", dict(mask_x=True), None, ("band", "band")),
             ("ddpm_oodad", dict(ood_AD=True), None, ("band", "band")),
             ("ddpm_ones_then_band", dict(mask_x=True), None, ("ones", "band")),
             ("ddim_maskx", dict(mask_x=True), 10, ("band", "band")),
             ("ddim_oodad", dict(ood_AD=True), 10, ("band", "band"))]


@pytest.mark.parametrize("tag,kw,S,seq", G14_CASES)
def test_consecutive_calls_carry_mask_x_like_the_reference(golden, tag, kw, S, seq):
    """G14: two sample() calls on ONE object (ddpm.py:780-781, 1023-1024, 1093-1117).  With {mask_x: True, ood_AD:
    False} the reference's second call leaves the OOD prediction unmasked; ood_AD re-arms the flag every call; the
    all-ones fallback clears it for later calls."""
    g = golden("g14_consecutive_calls")
    cond, masks = torch.from_numpy(g["cond"]), {"band": torch.from_numpy(g["band"]), "ones": torch.ones(2, 1, 32, 32)}
    o = diffusion_ref.SamplerOptions(timesteps=50, sampling_timesteps=S, branch_out=True, start_intermediate=True,
                                     start_timestep=2, data="mri", **kw)
    smp = diffusion_ref.RefSampler(diffusion_ref.make_model_fn(sd_of(CFG_MRI), CFG_MRI), o, 1, 32)
    for i, m in enumerate(seq):
        with torch.no_grad():
            out = smp.sample(cond, masks[m], (0.0, 2.0), 2, Noise(10)).numpy()
        np.testing.assert_allclose(out, g[f"{tag}_call{i + 1}"], atol=2e-5, rtol=0)


@pytest.mark.parametrize("tag,cfg,H,B", [("mnist28", CFG_MNIST, 28, 4), ("mri32", CFG_MRI, 32, 2)])
def test_training_forward_losses(golden, tag, cfg, H, B):
    """G15 (from the real reference): GaussianDiffusion.forward(train=False) and p_losses with explicit timesteps and
    offset noise, three objectives (ddpm.py:1156-1214).  The oracle's restatement on the fixture's draws."""
    g = golden("g15_p_losses")
    sd = {k: torch.from_numpy(v) for k, v in weights.procedural_state_dict(cfg, 0).items()}
    x0, cond = torch.from_numpy(g[tag + "_x0"]), torch.from_numpy(g[tag + "_cond"])
    for obj in ("pred_x0", "pred_noise", "pred_v"):
        smp = diffusion_ref.RefSampler(diffusion_ref.make_model_fn(sd, cfg), diffusion_ref.SamplerOptions(timesteps=100, objective=obj), 1, H)
        n0 = torch.from_numpy(rng.randn((B, 1, H, H), 10, 0))
        off = torch.from_numpy(rng.randn((B, 1), 10, 1))
        with torch.no_grad():
            l1, p1 = smp.p_losses(x0, cond, torch.from_numpy(g[f"{tag}_{obj}_t_fwd"]), n0)
            l2, p2 = smp.p_losses(x0, cond, torch.from_numpy(g[f"{tag}_{obj}_t_exp"]), n0, offset_noise=off, offset_noise_strength=0.1)
        for got, per, key in ((l1, p1, "fwd"), (l2, p2, "exp")):
            ref = float(g[f"{tag}_{obj}_loss_{key}"])
            assert abs(float(got) - ref) <= 2e-5 * max(1.0, abs(ref)), (obj, key, float(got), ref)
            assert np.allclose(per.numpy(), g[f"{tag}_{obj}_per_{key}"], rtol=2e-5, atol=1e-6)
    # the timesteps of forward(train=False): torch's generator seeded with 42 (ddpm.py:1210-1211), drawn on the host
    torch.random.manual_seed(42)
    assert torch.randint(0, 100, (B,)).tolist() == g[f"{tag}_pred_x0_t_fwd"].tolist()


def test_unet_constructor_options_fixture(golden):
    """Golden G17 (the real reference with learned_variance + learned Fourier time features, tools/make_goldens.py g17): the
    oracle reproduces it, the parameter inventory carries ``time_mlp.0.weights`` and a 17-wide first Linear, and the product's
    constructor accepts every option of the reference's (ddpm.py:294-300) -- self_condition too, whose failure is the
    forward's (tests/test_hip_unet.py)."""
    import localdiffusion_hallucination_amd as ldh
    g = golden("g17_unet_options")
    cfg = weights.UnetConfig(dim_mults=(1, 2, 4), full_attn=(False, False, True), mode="mnist", out_dim=2, learned_sinusoidal_dim=16)
    sd = {k: torch.from_numpy(v) for k, v in weights.procedural_state_dict(cfg, 0).items()}
    B, _, H, _ = [int(v) for v in g["shape"]]
    x = torch.from_numpy(rng.randn((B, 1, H, H), 17, 100))
    cond = torch.from_numpy(rng.uniform((B, 1, H, H), 17, 101, 0.0, 2.0))
    for t in (0, 7, 99):
        with torch.no_grad():
            y = unet_ref.unet_forward(sd, cfg, x, cond, torch.full((B,), t, dtype=torch.long))
        assert float((y - torch.from_numpy(g[f"t{t}_out"])).abs().max()) <= 1e-5
    net = ldh.Unet(dim=32, init_dim=32, dim_mults=(1, 2, 4), full_attn=(False, False, True), mode="mnist", learned_variance=True,
                   random_fourier_features=True, self_condition=True)
    assert net.cfg == cfg and net.out_dim == 2 and net.self_condition and net.random_or_learned_sinusoidal_cond
    assert list(net.state_dict().keys()) == list(sd.keys())
    assert ldh.Unet(dim=32, learned_variance=True, out_dim=5).out_dim == 5          # an explicit out_dim wins (ddpm.py:395)
