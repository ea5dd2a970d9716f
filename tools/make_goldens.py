#!/usr/bin/env python3
"""Golden-vector generator.  Runs ONLY in the build container (needs /root/reference).

It imports the *real* reference modules (``ddpm.py``, ``attend.py``, ``unet_model.py``) through
stub modules for the packages/files the reference imports but this image lacks,
loads this repo's procedural weights into the reference ``Unet`` by parameter name, replaces
``torch.randn`` / ``torch.randn_like`` with the portable counter-based generator
(localdiffusion_hallucination_amd.rng), runs the reference's own code on CPU and

  1. checks the oracle (oracle/unet_ref.py, oracle/diffusion_ref.py) against it, and
  2. writes small input/output fixtures to tests/golden/*.npz.

The fixtures are data (inputs + expected outputs); no reference source travels.
Usage:  python tools/make_goldens.py [--only G1,G2,...] [--skip-long]
"""
import argparse
import contextlib
import gzip
import io
import os
import sys
import tempfile
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")

import localdiffusion_hallucination_amd as ldh           # noqa: E402
from localdiffusion_hallucination_amd import rng, weights, schedule   # noqa: E402
from oracle import unet_ref, diffusion_ref                # noqa: E402


# ----------------------------------------------------------------------------- reference import
def import_reference():
    sys.dont_write_bytecode = True

    class _Dummy:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return self

    def mk(name):
        m = types.ModuleType(name)

        def ga(attr):
            if attr.startswith("__"):
                raise AttributeError(attr)
            return _Dummy
        m.__getattr__ = ga
        m.__path__ = []
        sys.modules[name] = m
    for n in ["torchvision", "torchvision.transforms", "torchvision.utils",
              "torchvision.transforms.functional", "ema_pytorch", "idx2numpy", "timm", "nibabel",
              "medpy", "medpy.io", "anomalib", "anomalib.models", "anomalib.models.components",
              "anomalib.models.patchcore", "anomalib.models.patchcore.anomaly_map",
              "anomalib.pre_processing", "anomalib.config", "train_fusion", "data", "models"]:
        mk(n)
    sys.path.insert(0, REF)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        import ddpm
    torch.Tensor.cuda = lambda self, *a, **k: self          # ddpm.py:743,748,754 on a CPU box
    return ddpm


class PortableNoise:
    """Monkeypatch target: every randn/randn_like call is draw #k of the portable stream."""

    def __init__(self, seed=10):
        self.seed, self.k = seed, 0

    def randn(self, *shape, **kw):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
            shape = tuple(shape[0])
        out = torch.from_numpy(rng.randn(tuple(shape), self.seed, self.k))
        self.k += 1
        return out

    def randn_like(self, x, **kw):
        return self.randn(tuple(x.shape))

    def __call__(self, shape):
        return self.randn(tuple(shape))


@contextlib.contextmanager
def reference_run(noise):
    """temp CWD with ./fusion_test, silenced stdout, portable noise patched into torch."""
    old = (torch.randn, torch.randn_like, os.getcwd())
    with tempfile.TemporaryDirectory() as d:
        os.makedirs(os.path.join(d, "fusion_test"))
        os.chdir(d)
        torch.randn, torch.randn_like = noise.randn, noise.randn_like
        try:
            with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
                yield
        finally:
            torch.randn, torch.randn_like = old[0], old[1]
            os.chdir(old[2])


def sd_torch(cfg, seed=0):
    return {k: torch.from_numpy(v) for k, v in weights.procedural_state_dict(cfg, seed).items()}


def build_reference_unet(ddpm, cfg, sd):
    m = ddpm.Unet(dim=cfg.dim, init_dim=cfg.init_dim, out_dim=cfg.out_dim, dim_mults=cfg.dim_mults,
                  channels=cfg.channels, full_attn=cfg.full_attn, mode=cfg.mode)
    ref_sd = m.state_dict()
    assert list(ref_sd.keys()) == list(sd.keys()), "parameter inventory differs from the reference"
    for k in ref_sd:
        assert tuple(ref_sd[k].shape) == tuple(sd[k].shape), k
    m.load_state_dict(sd)
    return m.eval()


def base_config(**kw):
    c = dict(branch_out=False, start_intermediate=False, start_timestep=2, data="mri", mask_x=False,
             mask_cond=False, ood_AD=False, ood_confidence=False, classifier=False,
             classifier_obj="tile", use_gt=False, use_gt_timestep=100, continue_fusion_timestep=0)
    c.update(kw)
    return c


def opts_from(config, T, S=None, objective="pred_x0", sched="sigmoid"):
    return diffusion_ref.SamplerOptions(
        timesteps=T, sampling_timesteps=S, objective=objective, beta_schedule=sched,
        branch_out=config["branch_out"], start_intermediate=config["start_intermediate"],
        start_timestep=config["start_timestep"], data=config["data"], mask_x=config["mask_x"],
        ood_AD=config["ood_AD"], ood_confidence=config["ood_confidence"], classifier=bool(config.get("classifier", False)),
        use_gt=bool(config.get("use_gt", False)), use_gt_timestep=int(config.get("use_gt_timestep", 100)))


def maxdiff(a, b):
    return float((a - b).abs().max())


def save(name, **arrs):
    os.makedirs(GOLD, exist_ok=True)
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print(f"  wrote {os.path.relpath(path, ROOT)}  ({os.path.getsize(path)/1024:.1f} KiB)")


CFG_MNIST = weights.UnetConfig(dim_mults=(1, 2, 4), full_attn=(False, False, True), mode="mnist")
CFG_MRI = weights.UnetConfig(mode="mri")
CFG_MVTEC = weights.UnetConfig(channels=3, out_dim=3, mode="mvtec")


def mnist_digits(n=4, label=3):
    """First n test digits with the given label (config.yaml:14 anomaly_name: 3)."""
    with gzip.open(os.path.join(REF, "MNIST/raw/t10k-images-idx3-ubyte.gz"), "rb") as f:
        raw = f.read()
    imgs = np.frombuffer(raw, dtype=np.uint8, offset=16).reshape(-1, 28, 28)
    with gzip.open(os.path.join(REF, "MNIST/raw/t10k-labels-idx1-ubyte.gz"), "rb") as f:
        lab = np.frombuffer(f.read(), dtype=np.uint8, offset=8)
    idx = np.nonzero(lab == label)[0][:n]
    return imgs[idx].copy()


def import_reference_dataset():
    """The reference's real ``data.py`` (its MNIST dataset class), loaded under another module name: the name
    ``data`` is a stub while ``ddpm`` is imported.  torchvision is absent here, so ``transforms.Compose`` -- the only
    torchvision symbol MNIST.__getitem__ reaches in test mode, with an empty list (data.py:792-795) -- is given its
    defining behaviour (apply the listed transforms in order)."""
    import importlib.util

    class Compose:
        def __init__(self, ts):
            self.ts = list(ts)

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x
    for n in ("torchvision.transforms",):
        sys.modules[n].Compose = Compose
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    # data.py:17,29 only takes a user-agent string from `datasets` (for an image downloader far from this path); the
    # installed package refuses to import beside the torchvision stub, so it is stubbed for the load as well
    saved = {n: sys.modules.get(n) for n in ("datasets", "datasets.utils", "datasets.utils.file_utils")}
    for n in saved:
        m = types.ModuleType(n)
        m.__path__ = []
        m.get_datasets_user_agent = lambda *a, **k: "stub"
        sys.modules[n] = m
    spec = importlib.util.spec_from_file_location("ref_data_real", os.path.join(REF, "data.py"))
    mod = importlib.util.module_from_spec(spec)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        try:
            spec.loader.exec_module(mod)
        finally:
            for n, m in saved.items():
                if m is None:
                    sys.modules.pop(n, None)
                else:
                    sys.modules[n] = m
    return mod


def mnist_lr_hr(digits, labels=None):
    """LR / HR pairs from the reference's OWN dataset code (MNIST.__getitem__, data.py:808-829), test mode.
    NB the reference indexes the [1,1,28,28] image with ``img[:, ::2, ::2]``, which decimates dims 0-2, i.e. only
    the ROWS (-> [1,1,14,28]) before the bilinear resize back to 28x28; the product's evalio.mnist_pairs follows
    that and is pinned by this fixture."""
    ref_data = import_reference_dataset()
    labels = np.full(len(digits), 3, dtype=np.uint8) if labels is None else labels
    ds = ref_data.MNIST({"augmentations": False}, digits, labels, train=False, num=[int(v) for v in set(labels.tolist())])
    assert len(ds) == len(digits)
    hr, lr = [], []
    for i in range(len(ds)):
        img, img_down, _ = ds[i]
        hr.append(img)
        lr.append(img_down)
    return torch.stack(lr), torch.stack(hr)


# ----------------------------------------------------------------------------- G1 schedules
def g1(ddpm):
    print("G1 schedule buffers")
    out = {}
    for sched in ("sigmoid", "linear", "cosine"):
        for T in (50, 100, 1000):
            gd = _ref_diffusion(ddpm, base_config(), _DummyModel(), 28, T, sched, "pred_x0", None)
            ours = schedule.make_buffers(T, sched, "pred_x0")
            orc = diffusion_ref.schedule_buffers(sched, T, "pred_x0")
            for name in schedule.BUFFER_NAMES:
                ref = getattr(gd, name)
                assert torch.equal(ref, ours[name]), (sched, T, name, "product schedule != reference")
                d = maxdiff(ref, orc[name]) / max(1e-30, float(ref.abs().max()))
                assert d < 2e-7, (sched, T, name, d)
                out[f"{sched}_{T}_{name}"] = ref.numpy()
    save("g1_schedules", **out)


class _DummyModel(torch.nn.Module):
    channels = 1
    out_dim = 1
    self_condition = False
    random_or_learned_sinusoidal_cond = False


def _ref_diffusion(ddpm, config, model, image_size, T, sched, objective, S):
    return ddpm.GaussianDiffusion(config, model, image_size=image_size, timesteps=T,
                                  beta_schedule=sched, objective=objective, auto_normalize=False,
                                  sampling_timesteps=S)


# ----------------------------------------------------------------------------- G2 forward
def tap_stats(t):
    t = t.detach().float()
    flat = t.flatten()
    idx = torch.linspace(0, flat.numel() - 1, 16).long()
    return np.concatenate([[float(t.mean()), float(t.norm())], flat[idx].numpy()]).astype(np.float32)


def g2(ddpm):
    print("G2 Unet.forward")
    out = {}
    cases = [("mnist28", CFG_MNIST, 4, 28, (0, 5, 99)),
             ("mri64", CFG_MRI, 1, 64, (0, 500, 999)),
             ("mvtec32", CFG_MVTEC, 2, 32, (3, 777))]
    for tag, cfg, B, H, ts in cases:
        sd = sd_torch(cfg)
        ref = build_reference_unet(ddpm, cfg, sd)
        cin = cfg.cond_in_channels
        x = torch.from_numpy(rng.randn((B, cfg.channels, H, H), 1, 100))
        cond = torch.from_numpy(rng.uniform((B, cin, H, H), 1, 101, 0.0, 2.0))
        for t in ts:
            tv = torch.full((B,), t, dtype=torch.long)
            hooks, caught = [], {}
            names = dict(ref.named_modules())
            want = [n for n in names if n in ("init_conv", "cond_model", "conv_fusion", "mid_block1",
                                              "mid_block2", "final_res_block")
                    or (n.count(".") == 2 and n.split(".")[0] in ("downs", "ups") and n[-1] in "01")]
            for n in want:
                hooks.append(names[n].register_forward_hook(
                    lambda mod, inp, o, n=n: caught.__setitem__(n, o.detach())))
            with torch.no_grad():
                y_ref = ref(x, cond, tv)
            for h in hooks:
                h.remove()
            taps = {}
            with torch.no_grad():
                y_orc = unet_ref.unet_forward(sd, cfg, x, cond, tv, taps)
            d = maxdiff(y_ref, y_orc)
            worst = max(maxdiff(caught[n], taps[n]) for n in want)
            print(f"  {tag} t={t}: oracle-vs-reference out {d:.2e}, worst tap {worst:.2e}")
            assert d <= 1e-5 and worst <= 1e-4, (tag, t, d, worst)
            out[f"{tag}_t{t}_out"] = y_ref.numpy()
            for n in want:
                out[f"{tag}_t{t}_tap_{n}"] = tap_stats(caught[n])
        out[f"{tag}_shape"] = np.array([B, cfg.channels, H, cin])
    save("g2_unet_forward", **out)


# ----------------------------------------------------------------------------- sampler runs
def run_pair(ddpm, cfg, config, T, S, B, H, cond, mask, lohi, objective="pred_x0", sched="sigmoid",
             record_ts=()):
    """Run reference sample() and oracle sample() on identical inputs -> (ref_out, orc_out, recs)."""
    sd = sd_torch(cfg)
    ref_model = build_reference_unet(ddpm, cfg, sd)
    config = dict(config)
    gd = _ref_diffusion(ddpm, config, ref_model, H, T, sched, objective, S).eval()
    noise = PortableNoise(10)
    t0 = time.time()
    with reference_run(noise):
        with torch.inference_mode():
            ref_out = gd.sample(cond.clone(), None, batch_size=B, mask=None if mask is None else mask.clone(),
                                min_max_val=lohi)
    t_ref = time.time() - t0
    return ref_out, t_ref, sd


def oracle_run(cfg, sd, config, T, S, B, H, cond, mask, lohi, objective="pred_x0", sched="sigmoid",
               record=None):
    o = opts_from(config, T, S, objective, sched)
    smp = diffusion_ref.RefSampler(diffusion_ref.make_model_fn(sd, cfg), o, cfg.channels, H)
    noise = PortableNoise(10)
    with torch.no_grad():
        if record is None:
            return smp.sample(cond, mask, lohi, B, noise)
        branch, fuse = o.branch_out, o.start_intermediate
        mask_x = o.mask_x or o.ood_AD or o.ood_confidence
        return smp.p_sample_loop(cond, mask, lohi, (B, cfg.channels, H, H), noise, branch, fuse, mask_x,
                                 record=record)


def to_np(x):
    if isinstance(x, (list, tuple)):
        return np.stack([t.numpy() for t in x])
    return x.numpy()


def compare(tag, ref_out, orc_out, tol):
    a, b = torch.from_numpy(to_np(ref_out)), torch.from_numpy(to_np(orc_out))
    assert a.shape == b.shape, (tag, a.shape, b.shape)
    d = maxdiff(a, b)
    print(f"  {tag}: shape {tuple(a.shape)} oracle-vs-reference max-abs {d:.3e}")
    assert d <= tol, (tag, d)
    return d


def band_mask(B, H, ncols):
    m = torch.zeros(B, 1, H, H)
    m[:, :, :, :ncols] = 1.0
    return m


def g3(ddpm):
    """Single p_sample steps (non-branch, branch, fusion) are covered as T=3 runs with
    start_timestep=1: step t=2 is a pure branch step, t=1 the fusion step, t=0 a joint step."""
    print("G3 three-step runs (branch / fusion / joint)")
    cfg, H, B = CFG_MNIST, 28, 2
    cond = torch.from_numpy(rng.uniform((B, 1, H, H), 3, 1, 0.0, 2.0))
    mask = band_mask(B, H, 7)
    out = {"cond": cond.numpy(), "mask": mask.numpy()}
    for tag, kw in [("nonbranch", dict(data="mnist")),
                    ("branch_fuse_mnist", dict(data="mnist", branch_out=True, start_intermediate=True,
                                               start_timestep=1, mask_x=True)),
                    ("branch_fuse_mri", dict(data="mri", branch_out=True, start_intermediate=True,
                                             start_timestep=1, mask_x=True)),
                    ("branch_nofuse", dict(data="mri", branch_out=True, start_intermediate=False,
                                           mask_x=True))]:
        config = base_config(**kw)
        ref_out, _, sd = run_pair(ddpm, cfg, config, 3, None, B, H, cond, mask, (0.0, 2.0))
        orc = oracle_run(cfg, sd, base_config(**kw), 3, None, B, H, cond, mask, (0.0, 2.0))
        compare("G3 " + tag, ref_out, orc, 1e-5)
        out[tag] = to_np(ref_out)
    save("g3_three_step", **out)


def g4(ddpm):
    print("G4 cfg1: MNIST 28x28, T=100, 4 patches, branch + fusion (config.yaml defaults)")
    cfg, H, B, T = CFG_MNIST, 28, 4, 100
    digits = mnist_digits(B, 3)
    lr, hr = mnist_lr_hr(digits)
    mask = band_mask(B, H, 7)                                   # test.py:379-381
    kw = dict(data="mnist", branch_out=True, start_intermediate=True, start_timestep=2, mask_x=True,
              ood_AD=True)
    ref_out, t_ref, sd = run_pair(ddpm, cfg, base_config(**kw), T, None, B, H, lr, mask, (0.0, 2.0))
    recs = {}
    keep = (99, 50, 10, 2, 1, 0)
    orc = oracle_run(cfg, sd, base_config(**kw), T, None, B, H, lr, mask, (0.0, 2.0),
                     record=lambda t, x: recs.__setitem__(t, to_np(x)) if t in keep else None)
    compare("G4", ref_out, orc, 2e-5)
    print(f"  reference sample() wall {t_ref:.1f}s")
    save("g4_cfg1_mnist", digits=digits, cond=lr.numpy(), hr=hr.numpy(), mask=mask.numpy(),
         final=to_np(ref_out), **{f"x_after_t{t}": v for t, v in recs.items()})


def g5(ddpm):
    print("G5 cfg2: 4-stage 1-ch 128x128, T=1000, fp32, non-branch  (several minutes)")
    cfg, H, B, T = CFG_MRI, 128, 1, 1000
    cond = torch.from_numpy(rng.uniform((B, 1, H, H), 5, 1, 0.0, 2.0))
    kw = dict(data="mri")
    ref_out, t_ref, sd = run_pair(ddpm, cfg, base_config(**kw), T, None, B, H, cond, None, (0.0, 2.0))
    print(f"  reference sample() wall {t_ref:.1f}s")
    recs = {}
    keep = (999, 750, 500, 250, 100, 10, 0)
    t0 = time.time()
    orc = oracle_run(cfg, sd, base_config(**kw), T, None, B, H, cond, None, (0.0, 2.0),
                     record=lambda t, x: recs.__setitem__(t, to_np(x)) if t in keep else None)
    print(f"  oracle sample() wall {time.time()-t0:.1f}s")
    compare("G5", ref_out, orc, 1e-4)
    save("g5_cfg2_mri128", final=to_np(ref_out), ref_seconds=np.array([t_ref]),
         **{f"x_after_t{t}": v for t, v in recs.items()})


def g6(ddpm):
    print("G6 branch+fusion T=50: data='mri' 32x32 (UNet out kept) and data='mnist' 28x28")
    out = {}
    for tag, cfg, H, data in [("mri32", CFG_MRI, 32, "mri"), ("mnist28", CFG_MNIST, 28, "mnist")]:
        B, T = 2, 50
        cond = torch.from_numpy(rng.uniform((B, 1, H, H), 6, 1, 0.0, 2.0))
        mask = band_mask(B, H, H // 4)
        kw = dict(data=data, branch_out=True, start_intermediate=True, start_timestep=2, mask_x=True)
        ref_out, _, sd = run_pair(ddpm, cfg, base_config(**kw), T, None, B, H, cond, mask, (0.0, 2.0))
        orc = oracle_run(cfg, sd, base_config(**kw), T, None, B, H, cond, mask, (0.0, 2.0))
        compare("G6 " + tag, ref_out, orc, 2e-5)
        out[tag + "_final"] = to_np(ref_out)
    save("g6_branch_fusion", **out)


def g7(ddpm):
    print("G7 DDIM S=50 of T=1000, 64x64 1-ch, branch + fusion; and S=10 of T=50 no-fusion (list)")
    cfg, H, B = CFG_MRI, 64, 1
    cond = torch.from_numpy(rng.uniform((B, 1, H, H), 7, 1, 0.0, 2.0))
    mask = torch.zeros(B, 1, H, H)
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(H), indexing="ij")
    mask[:, :, ((yy - H // 2) ** 2 + (xx - H // 2) ** 2) <= (H // 8) ** 2] = 1.0
    out = {"mask": mask.numpy()}
    kw = dict(data="mri", branch_out=True, start_intermediate=True, start_timestep=2, mask_x=True)
    ref_out, _, sd = run_pair(ddpm, cfg, base_config(**kw), 1000, 50, B, H, cond, mask, (0.0, 2.0))
    orc = oracle_run(cfg, sd, base_config(**kw), 1000, 50, B, H, cond, mask, (0.0, 2.0))
    compare("G7 fused", ref_out, orc, 2e-5)
    out["fused_final"] = to_np(ref_out)
    kw = dict(data="mri", branch_out=True, start_intermediate=False, mask_x=True)
    ref_out, _, sd = run_pair(ddpm, cfg, base_config(**kw), 50, 10, B, H, cond, mask, (0.0, 2.0))
    orc = oracle_run(cfg, sd, base_config(**kw), 50, 10, B, H, cond, mask, (0.0, 2.0))
    compare("G7 nofuse", ref_out, orc, 2e-5)
    out["nofuse_final"] = to_np(ref_out)
    kw = dict(data="mri")
    ref_out, _, sd = run_pair(ddpm, cfg, base_config(**kw), 50, 10, B, H, cond, None, (0.0, 2.0))
    orc = oracle_run(cfg, sd, base_config(**kw), 50, 10, B, H, cond, None, (0.0, 2.0))
    compare("G7 single", ref_out, orc, 2e-5)
    out["single_final"] = to_np(ref_out)
    save("g7_ddim", **out)


def g8(ddpm):
    print("G8 all-ones mask fallback + pred_noise / pred_v objectives (non-branch)")
    cfg, H, B, T = CFG_MNIST, 28, 2, 20
    cond = torch.from_numpy(rng.uniform((B, 1, H, H), 8, 1, 0.0, 2.0))
    out = {}
    kw = dict(data="mnist", branch_out=True, start_intermediate=True, start_timestep=2, mask_x=True)
    ones = torch.ones(B, 1, H, H)
    ref_out, _, sd = run_pair(ddpm, cfg, base_config(**kw), T, None, B, H, cond, ones, (0.0, 2.0))
    orc = oracle_run(cfg, sd, base_config(**kw), T, None, B, H, cond, ones, (0.0, 2.0))
    compare("G8 all-ones", ref_out, orc, 1e-5)
    out["allones_final"] = to_np(ref_out)
    for obj in ("pred_noise", "pred_v"):
        kw = dict(data="mnist")
        ref_out, _, sd = run_pair(ddpm, cfg, base_config(**kw), T, None, B, H, cond, None, (0.0, 2.0),
                                  objective=obj)
        orc = oracle_run(cfg, sd, base_config(**kw), T, None, B, H, cond, None, (0.0, 2.0), objective=obj)
        compare("G8 " + obj, ref_out, orc, 1e-4)
        out[obj + "_final"] = to_np(ref_out)
    save("g8_fallback_objectives", **out)


class StubClassifier:
    """Deterministic stand-in for Classifier_PatchCore (ddpm.py:622-625, anomalib is not installed): same call
    convention -- x0 -> (score, _, _) -- rejecting (score -1) the first ``reject`` calls, then accepting (+1).
    Records the x0 it was shown."""
    def __init__(self, reject):
        self.reject, self.calls, self.seen = reject, 0, []

    def __call__(self, x0):
        self.calls += 1
        self.seen.append(x0.detach().clone().cpu())
        return (torch.tensor(-1.0 if self.calls <= self.reject else 1.0), None, None)


def g9(ddpm):
    print("G9 classifier-gated re-branching (fusion(), ddpm.py:883-916) with a stub classifier")
    out = {}
    for tag, cfg, H, data, reject in [("mnist28_reject2", CFG_MNIST, 28, "mnist", 2), ("mri32_reject3", CFG_MRI, 32, "mri", 3),
                                      ("mri32_reject_all", CFG_MRI, 32, "mri", 99)]:
        B, T = 2, 12
        cond = torch.from_numpy(rng.uniform((B, 1, H, H), 9, 1, 0.0, 2.0))
        mask = band_mask(B, H, H // 4)
        kw = dict(data=data, branch_out=True, start_intermediate=True, start_timestep=7, mask_x=True, classifier=True)
        sd = sd_torch(cfg)
        ref_model = build_reference_unet(ddpm, cfg, sd)
        gd = _ref_diffusion(ddpm, base_config(**kw), ref_model, H, T, "sigmoid", "pred_x0", None).eval()
        stub = StubClassifier(reject)
        gd.classifier = stub
        noise = PortableNoise(10)
        with reference_run(noise):
            with torch.inference_mode():
                ref_out = gd.sample(cond.clone(), None, batch_size=B, mask=mask.clone(), min_max_val=(0.0, 2.0))
        o = opts_from(base_config(**kw), T)
        smp = diffusion_ref.RefSampler(diffusion_ref.make_model_fn(sd, cfg), o, cfg.channels, H)
        stub2 = StubClassifier(reject)
        smp.classifier = stub2
        with torch.no_grad():
            orc = smp.sample(cond, mask, (0.0, 2.0), B, PortableNoise(10))
        compare("G9 " + tag, ref_out, orc, 2e-5)
        assert stub.calls == stub2.calls, (stub.calls, stub2.calls)
        for a, b in zip(stub.seen, stub2.seen):
            assert maxdiff(a, b) <= 2e-5
        print(f"    classifier calls: {stub.calls}")
        out[tag + "_final"] = to_np(ref_out)
        out[tag + "_calls"] = np.array([stub.calls, reject])
        if tag == "mnist28_reject2":
            out["cond28"], out["mask28"] = cond.numpy(), mask.numpy()
        elif tag == "mri32_reject3":
            out["cond32"], out["mask32"] = cond.numpy(), mask.numpy()
    save("g9_classifier_gate", **out)


def g10(ddpm):
    print("G10 use_gt start (ddpm.py:937-944) and return_all_timesteps / return_all_outputs (:946,959-977,1072)")
    out = {}

    def both(cfg, H, config, T, S, B, cond, mask, hr, rat, rao):
        sd = sd_torch(cfg)
        ref_model = build_reference_unet(ddpm, cfg, sd)
        gd = _ref_diffusion(ddpm, dict(config), ref_model, H, T, "sigmoid", "pred_x0", S).eval()
        with reference_run(PortableNoise(10)):
            with torch.inference_mode():
                ref = gd.sample(cond.clone(), None if hr is None else hr.clone(), batch_size=B,
                                mask=None if mask is None else mask.clone(), min_max_val=(0.0, 2.0),
                                return_all_timesteps=rat, return_all_outputs=rao)
        o = opts_from(config, T, S)
        smp = diffusion_ref.RefSampler(diffusion_ref.make_model_fn(sd, cfg), o, cfg.channels, H)
        with torch.no_grad():
            orc = smp.sample(cond, mask, (0.0, 2.0), B, PortableNoise(10), gt=hr, return_all_timesteps=rat,
                             return_all_outputs=rao)
        return ref, orc

    def flat_x0(lst):
        """x_start_lst -> one array [steps, (2,) B, C, H, W] per homogeneous run of entries"""
        pairs = [np.stack([e[0].numpy(), e[1].numpy()]) for e in lst if isinstance(e, (list, tuple))]
        singles = [e.numpy() for e in lst if not isinstance(e, (list, tuple))]
        return (np.stack(pairs) if pairs else np.zeros((0,), np.float32)), (np.stack(singles) if singles else np.zeros((0,), np.float32))

    # (a) single-branch, use_gt start at t0 = 20 of T = 50, every x_t and every x0 returned
    B, H, T = 2, 28, 50
    cond = torch.from_numpy(rng.uniform((B, 1, H, H), 10, 1, 0.0, 2.0))
    hr = torch.from_numpy(rng.uniform((B, 1, H, H), 10, 2, 0.0, 2.0))
    kw = base_config(data="mnist", start_intermediate=True, use_gt=True, use_gt_timestep=20)
    (r_ret, r_x0, r_cm), (o_ret, o_x0, o_cm) = both(CFG_MNIST, H, kw, T, None, B, cond, None, hr, True, True)
    compare("G10a x_t history", r_ret, o_ret, 2e-5)
    assert tuple(r_ret.shape) == (B, 21, 1, H, H) and len(r_x0) == len(o_x0) == 20 and r_cm == [] and o_cm == []
    for a, b in zip(r_x0, o_x0):
        assert maxdiff(a, b) <= 2e-5
    out["a_cond"], out["a_hr"], out["a_hist"] = cond.numpy(), hr.numpy(), to_np(r_ret)
    out["a_x0"] = np.stack([e.numpy() for e in r_x0])
    # (b) branch + fusion from a use_gt start, x0 history (pairs for the branch steps, then fused / joint x0)
    B, H, T = 2, 32, 40
    cond = torch.from_numpy(rng.uniform((B, 1, H, H), 10, 3, 0.0, 2.0))
    hr = torch.from_numpy(rng.uniform((B, 1, H, H), 10, 4, 0.0, 2.0))
    mask = band_mask(B, H, H // 4)
    kw = base_config(data="mri", branch_out=True, start_intermediate=True, start_timestep=3, mask_x=True, use_gt=True,
                     use_gt_timestep=12)
    (r_ret, r_x0, _), (o_ret, o_x0, _) = both(CFG_MRI, H, kw, T, None, B, cond, mask, hr, False, True)
    compare("G10b final", r_ret, o_ret, 2e-5)
    assert len(r_x0) == len(o_x0) == 12
    for a, b in zip(r_x0, o_x0):
        assert isinstance(a, (list, tuple)) == isinstance(b, (list, tuple))
        assert maxdiff(torch.from_numpy(to_np(a)), torch.from_numpy(to_np(b))) <= 2e-5
    out["b_cond"], out["b_hr"], out["b_mask"], out["b_final"] = cond.numpy(), hr.numpy(), mask.numpy(), to_np(r_ret)
    out["b_x0_pairs"], out["b_x0_single"] = flat_x0(r_x0)
    assert out["b_x0_pairs"].shape[0] == 8 and out["b_x0_single"].shape[0] == 4      # t = 11..4 branch, 3 fusion, 2..0 joint
    # (c) DDIM, single branch, every x_t returned
    B, H, T, S = 2, 28, 50, 10
    cond = torch.from_numpy(rng.uniform((B, 1, H, H), 10, 5, 0.0, 2.0))
    kw = base_config(data="mnist")
    r_ret, o_ret = both(CFG_MNIST, H, kw, T, S, B, cond, None, None, True, False)
    compare("G10c DDIM x_t history", r_ret, o_ret, 2e-5)
    assert tuple(r_ret.shape) == (B, S + 1, 1, H, H)
    out["c_cond"], out["c_hist"] = cond.numpy(), to_np(r_ret)
    # (d) return_all_timesteps with branches: the reference's torch.stack raises; record the exception type
    kw = base_config(data="mri", branch_out=True, start_intermediate=True, start_timestep=3, mask_x=True)
    for S_ in (None, 10):
        try:
            both(CFG_MRI, 32, kw, 40, S_, 2, torch.from_numpy(rng.uniform((2, 1, 32, 32), 10, 3, 0.0, 2.0)), band_mask(2, 32, 8), None, True, False)
            raise AssertionError("expected the reference to fail")
        except TypeError as e:
            print(f"    branch + return_all_timesteps (S={S_}): reference raises TypeError ({str(e)[:60]}...)")
    save("g10_use_gt_return_all", **out)


def g11(ddpm):
    """The floor of ANY 16-bit-storage implementation on cfg2: the reference's own fp32 code with nothing but its
    denoiser OUTPUT rounded to bf16 / fp16 once per step (an implementation that stores activations in 16 bits rounds
    hundreds of tensors per step).  The distance of these runs from the fp32 golden G5 is the yardstick the
    16-bit chain tests price the HIP path against (tests/test_hip_lowp_chain.py)."""
    print("G11 cfg2 with the reference's denoiser output rounded to 16 bits once per step  (several minutes)")
    cfg, H, B, T = CFG_MRI, 128, 1, 1000
    cond = torch.from_numpy(rng.uniform((B, 1, H, H), 5, 1, 0.0, 2.0))
    g5_ = np.load(os.path.join(GOLD, "g5_cfg2_mri128.npz"))
    out = {}
    keep = (999, 750, 500, 250, 100, 10, 0)
    for tag, dt in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
        sd = sd_torch(cfg)
        ref_model = build_reference_unet(ddpm, cfg, sd)
        inner = ref_model.forward
        ref_model.forward = lambda *a, _f=inner, _dt=dt, **k: _f(*a, **k).to(_dt).to(torch.float32)
        gd = _ref_diffusion(ddpm, base_config(data="mri"), ref_model, H, T, "sigmoid", "pred_x0", None).eval()
        t0 = time.time()
        with reference_run(PortableNoise(10)):
            with torch.inference_mode():
                hist = gd.sample(cond.clone(), None, batch_size=B, mask=None, min_max_val=(0.0, 2.0), return_all_timesteps=True)
        hist = hist.numpy()
        print(f"  {tag}: reference sample() wall {time.time()-t0:.1f}s")
        for t in keep:
            d = np.abs(hist[:, T - t] - g5_[f"x_after_t{t}"])
            print(f"    output rounded to {tag}: x after t={t}: max-abs {d.max():.3e} mean-abs {d.mean():.3e}")
            out[f"{tag}_x_after_t{t}"] = hist[:, T - t]
    save("g11_cfg2_output_rounded", **out)


def g12(ddpm):
    """Checkpoint format pinned to what the reference's own Trainer.save writes (ddpm.py:1495-1507) and
    Trainer.load reads (:1509-1527).  Both methods are called unbound on a minimal trainer object holding a real
    accelerate.Accelerator, the reference GaussianDiffusion, an Adam optimiser and an EMA wrapper (ema_pytorch is
    absent from this image: the stand-in has the package's state_dict layout -- online_model.*, ema_model.*,
    initted, step).  The file is ~40 MB of procedural weights, so what is committed is its MANIFEST (every key with
    shape and dtype); tests rebuild the file with checkpoint.save_reference_checkpoint and compare manifests."""
    import copy
    import json
    import pathlib
    from accelerate import Accelerator
    from localdiffusion_hallucination_amd import checkpoint
    print("G12 Trainer.save / Trainer.load checkpoint format")
    cfg, H, T = CFG_MNIST, 28, 100
    sd = sd_torch(cfg, 3)
    ref_model = build_reference_unet(ddpm, cfg, sd)
    kw = base_config(data="mnist", branch_out=True, start_intermediate=True, start_timestep=2, mask_x=True)
    gd = _ref_diffusion(ddpm, kw, ref_model, H, T, "sigmoid", "pred_x0", None)

    class EMA(torch.nn.Module):
        def __init__(self, model):
            super().__init__()
            self.online_model = model
            self.ema_model = copy.deepcopy(model)
            self.register_buffer("initted", torch.tensor(True))
            self.register_buffer("step", torch.tensor(0))

    def manifest(data):
        def walk(d):
            return {k: ([list(v.shape), str(v.dtype)] if torch.is_tensor(v) else walk(v) if isinstance(v, dict) else type(v).__name__)
                    for k, v in d.items()}
        m = {k: (walk(v) if isinstance(v, dict) else type(v).__name__) for k, v in data.items()}
        m["opt"] = "dict"                      # optimiser state is training-side, not read by the sampling path
        return m

    with tempfile.TemporaryDirectory() as d:
        tr = types.SimpleNamespace(accelerator=Accelerator(cpu=True), model=gd, opt=torch.optim.Adam(gd.parameters(), lr=1e-4),
                                   ema=EMA(gd), step=2900, results_folder=pathlib.Path(d))
        ddpm.Trainer.save(tr, "best2900")                     # the reference's own writer
        path = os.path.join(d, "model-best2900.pt")
        data = torch.load(path, map_location="cpu", weights_only=True)      # loads under the safe unpickler
        man = manifest(data)
        # the product reads what the reference wrote ...
        net = ldh.Unet(dim=32, init_dim=32, dim_mults=cfg.dim_mults, full_attn=cfg.full_attn, mode=cfg.mode)
        mine = ldh.GaussianDiffusion(dict(kw), net, image_size=H, timesteps=T, objective="pred_x0")
        info = checkpoint.load_reference_checkpoint(path, mine)
        assert info == {"step": 2900, "source": "ema", "missing": [], "unexpected": []}, info
        for k, v in gd.state_dict().items():
            assert torch.equal(mine.state_dict()[k], v), k
        # ... and the reference reads what the product writes (Trainer.load on a second trainer object)
        path2 = os.path.join(d, "model-mine.pt")
        checkpoint.save_reference_checkpoint(mine, path2, step=2900)
        assert manifest(torch.load(path2, map_location="cpu", weights_only=True)) == man
        gd2 = _ref_diffusion(ddpm, kw, build_reference_unet(ddpm, cfg, sd_torch(cfg, 4)), H, T, "sigmoid", "pred_x0", None)
        tr2 = types.SimpleNamespace(accelerator=Accelerator(cpu=True), model=gd2, opt=torch.optim.Adam(gd2.parameters(), lr=1e-4),
                                    ema=EMA(gd2), step=0, results_folder=pathlib.Path(d))
        with contextlib.redirect_stdout(io.StringIO()):
            ddpm.Trainer.load(tr2, "mine")
        assert tr2.step == 2900
        for k, v in gd.state_dict().items():
            assert torch.equal(gd2.state_dict()[k], v), k
    with open(os.path.join(GOLD, "g12_trainer_save_manifest.json"), "w") as f:
        json.dump(man, f, indent=0, sort_keys=True)
    print(f"  wrote tests/golden/g12_trainer_save_manifest.json ({len(man['model'])} model keys, {len(man['ema'])} ema keys)")


def g13(ddpm):
    """How far the REFERENCE is from ITSELF on cfg2 when only its summation order changes: the same code, weights and
    noise with torch.set_num_threads(1) instead of all cores (MKL-DNN / ATen split reductions differently).  This is the
    reproducibility floor of the 1e-3 gate: any fp32 implementation that sums in another order sits at this distance."""
    print("G13 cfg2 on ONE thread vs the golden made on all cores  (tens of minutes)")
    cfg, H, B, T = CFG_MRI, 128, 1, 1000
    cond = torch.from_numpy(rng.uniform((B, 1, H, H), 5, 1, 0.0, 2.0))
    g5_ = np.load(os.path.join(GOLD, "g5_cfg2_mri128.npz"))
    sd = sd_torch(cfg)
    ref_model = build_reference_unet(ddpm, cfg, sd)
    gd = _ref_diffusion(ddpm, base_config(data="mri"), ref_model, H, T, "sigmoid", "pred_x0", None).eval()
    n0 = torch.get_num_threads()
    torch.set_num_threads(1)
    t0 = time.time()
    try:
        with reference_run(PortableNoise(10)):
            with torch.inference_mode():
                hist = gd.sample(cond.clone(), None, batch_size=B, mask=None, min_max_val=(0.0, 2.0), return_all_timesteps=True).numpy()
    finally:
        torch.set_num_threads(n0)
    print(f"  reference sample() on 1 thread: wall {time.time()-t0:.1f}s")
    out = {}
    for t in (999, 750, 500, 250, 100, 10, 0):
        d = np.abs(hist[:, T - t] - g5_[f"x_after_t{t}"])
        print(f"    1 thread vs {n0} threads: x after t={t}: max-abs {d.max():.3e} mean-abs {d.mean():.3e}")
        out[f"maxabs_t{t}"] = np.float64(d.max())
        out[f"meanabs_t{t}"] = np.float64(d.mean())
    out["final_1thread"] = hist[:, -1]
    out["threads"] = np.array([1, n0])
    save("g13_cfg2_reference_self_distance", **out)


G16_GAIN = 0.25


def g16(ddpm):
    """The end-to-end pin of the 16-bit storage modes: cfg2's shape and schedule (128x128 1-ch, T = 1000,
    fp32, non-branch) run by the REAL reference on a CONTRACTIVE procedural denoiser -- the same weights with final_conv's
    gain 0.25 instead of 3 (weights.procedural_state_dict(final_gain=...)).  With gain 3 the last 100 steps of the chain
    amplify any perturbation x40-170 (G11 / G13: a random-init net's prediction swings over the whole range whatever x_t
    is), so the final-image distance of a 16-bit run says nothing about the implementation; with gain 0.25 the same
    perturbation (weights rounded to bf16) grows x1.7 in the mean from t = 100 to t = 0 and not at all over the last 10
    steps (tools/exp_contractive.py), as for a trained denoiser whose prediction at small t stays near x_t.  Stored: the
    states after t = 999, 750, 500, 250, 100, 10, 0, the oracle checked bit for bit, and the reference's own 1-thread
    self-distance on this net (the floor any other summation order has)."""
    print("G16 cfg2 shape on the contractive procedural net (final_conv gain 0.25)  (tens of minutes)")
    cfg, H, B, T = CFG_MRI, 128, 1, 1000
    cond = torch.from_numpy(rng.uniform((B, 1, H, H), 5, 1, 0.0, 2.0))
    sd = {k: torch.from_numpy(v) for k, v in weights.procedural_state_dict(cfg, 0, final_gain=G16_GAIN).items()}
    keep = (999, 750, 500, 250, 100, 10, 0)
    out = {}

    def ref_hist(threads):
        ref_model = build_reference_unet(ddpm, cfg, sd)
        gd = _ref_diffusion(ddpm, base_config(data="mri"), ref_model, H, T, "sigmoid", "pred_x0", None).eval()
        n0 = torch.get_num_threads()
        torch.set_num_threads(threads)
        t0 = time.time()
        try:
            with reference_run(PortableNoise(10)):
                with torch.inference_mode():
                    h = gd.sample(cond.clone(), None, batch_size=B, mask=None, min_max_val=(0.0, 2.0), return_all_timesteps=True).numpy()
        finally:
            torch.set_num_threads(n0)
        print(f"  reference sample() on {threads} threads: wall {time.time()-t0:.1f}s")
        return h
    hist = ref_hist(os.cpu_count())
    recs = {}
    orc = oracle_run(cfg, sd, base_config(data="mri"), T, None, B, H, cond, None, (0.0, 2.0),
                     record=lambda t, x: recs.__setitem__(t, to_np(x)) if t in keep else None)
    compare("G16", torch.from_numpy(hist[:, -1]), orc, 1e-4)
    for t in keep:
        assert maxdiff(torch.from_numpy(hist[:, T - t]), torch.from_numpy(recs[t])) == 0.0, t
        out[f"x_after_t{t}"] = hist[:, T - t]
    out["final"] = hist[:, -1]
    h1 = ref_hist(1)
    for t in keep:
        d = np.abs(h1[:, T - t] - hist[:, T - t])
        print(f"    1 thread vs {os.cpu_count()} threads: x after t={t}: max-abs {d.max():.3e} mean-abs {d.mean():.3e}")
        out[f"self_maxabs_t{t}"] = np.float64(d.max())
        out[f"self_meanabs_t{t}"] = np.float64(d.mean())
    out["final_gain"] = np.float64(G16_GAIN)
    print(f"  x_0 range [{hist[:, -1].min():.3f}, {hist[:, -1].max():.3f}] std {hist[:, -1].std():.4f}")
    save("g16_cfg2_contractive", **out)


def g14(ddpm):
    """Two (or three) consecutive sample() calls on ONE GaussianDiffusion object.  The reference clears
    config['mask_x'] at the fusion step (ddpm.py:780-781, 1023-1024) and in the all-ones fallback (:1114) and
    re-arms it in sample() only under ood_AD / ood_confidence (:1106-1108), so with {mask_x: True, ood_AD: False}
    the SECOND call runs its branch steps without masking the OOD prediction.  Every call gets a fresh portable
    noise stream (DDPM re-seeds per call, :934; for DDIM that is this fixture's convention)."""
    print("G14 consecutive sample() calls on one object (mask_x call state), DDPM and DDIM")
    cfg, H, B = CFG_MRI, 32, 2
    cond = torch.from_numpy(rng.uniform((B, 1, H, H), 14, 1, 0.0, 2.0))
    band, ones = band_mask(B, H, H // 4), torch.ones(B, 1, H, H)
    out = {"cond": cond.numpy(), "band": band.numpy()}
    fuse = dict(data="mri", branch_out=True, start_intermediate=True, start_timestep=2)
    cases = [("ddpm_maskx", dict(fuse, mask_x=True), 50, None, [band, band]),
             ("ddpm_oodad", dict(fuse, ood_AD=True), 50, None, [band, band]),
             ("ddpm_ones_then_band", dict(fuse, mask_x=True), 50, None, [ones, band]),
             ("ddim_maskx", dict(fuse, mask_x=True), 50, 10, [band, band]),
             ("ddim_oodad", dict(fuse, ood_AD=True), 50, 10, [band, band])]
    for tag, kw, T, S, masks in cases:
        sd = sd_torch(cfg)
        gd = _ref_diffusion(ddpm, base_config(**kw), build_reference_unet(ddpm, cfg, sd), H, T, "sigmoid", "pred_x0", S).eval()
        smp = diffusion_ref.RefSampler(diffusion_ref.make_model_fn(sd, cfg), opts_from(base_config(**kw), T, S), cfg.channels, H)
        refs = []
        for i, m in enumerate(masks):
            with reference_run(PortableNoise(10)):
                with torch.inference_mode():
                    r = gd.sample(cond.clone(), None, batch_size=B, mask=m.clone(), min_max_val=(0.0, 2.0))
            with torch.no_grad():
                o = smp.sample(cond, m, (0.0, 2.0), B, PortableNoise(10))
            compare(f"G14 {tag} call {i + 1}", r, o, 2e-5)
            refs.append(to_np(r))
            out[f"{tag}_call{i + 1}"] = to_np(r)
        d = float(np.abs(refs[-1] - refs[0]).max()) if refs[-1].shape == refs[0].shape else float("nan")
        print(f"    last call vs first call: max-abs {d:.3e}")
        if tag.endswith("_maskx"):
            assert d > 1e-3, "the second call should differ (mask_x is disarmed by the first call's fusion step)"
        if tag.endswith("_oodad"):
            assert d == 0.0, "ood_AD re-arms mask_x: both calls must agree"
    save("g14_consecutive_calls", **out)


def g15(ddpm):
    print("G15 training-side forward: GaussianDiffusion.forward / p_losses (ddpm.py:1156-1214), three objectives")
    out = {}
    for tag, cfg, H, B in (("mnist28", CFG_MNIST, 28, 4), ("mri32", CFG_MRI, 32, 2)):
        sd = sd_torch(cfg)
        x0 = torch.from_numpy(rng.uniform((B, 1, H, H), 15, 1, 0.0, 2.0))
        cond = torch.from_numpy(rng.uniform((B, 1, H, H), 15, 2, 0.0, 2.0))
        out[tag + "_x0"], out[tag + "_cond"] = to_np(x0), to_np(cond)
        for obj in ("pred_x0", "pred_noise", "pred_v"):
            T = 100
            ref_model = build_reference_unet(ddpm, cfg, sd)
            gd = _ref_diffusion(ddpm, base_config(data="mnist" if cfg is CFG_MNIST else "mri"), ref_model, H, T, "sigmoid", obj, None).eval()
            # (a) forward(img, cond, train=False): the reference seeds torch's generator with 42 and draws t with torch.randint
            noise = PortableNoise(10)
            with reference_run(noise):
                with torch.inference_mode():
                    loss_fwd = gd(x0.clone(), cond.clone(), False)
            torch.random.manual_seed(42)
            t_fwd = torch.randint(0, T, (B,)).long()
            # (b) p_losses with explicit timesteps (first / middle / last) and offset noise
            t_exp = torch.tensor(([0, T // 2, T - 1, 7] * B)[:B])
            noise = PortableNoise(10)
            with reference_run(noise):
                with torch.inference_mode():
                    loss_exp = gd.p_losses(x0.clone(), cond.clone(), t_exp, offset_noise_strength=0.1)
            # oracle restatement on the same draws
            o = opts_from(base_config(), T, None, obj)
            smp = diffusion_ref.RefSampler(diffusion_ref.make_model_fn(sd, cfg), o, 1, H)
            pn = PortableNoise(10)
            with torch.no_grad():
                o_fwd, per_fwd = smp.p_losses(x0, cond, t_fwd, pn.randn_like(x0))
                pn = PortableNoise(10)
                n1 = pn.randn_like(x0)
                off = pn.randn(B, 1)
                o_exp, per_exp = smp.p_losses(x0, cond, t_exp, n1, offset_noise=off, offset_noise_strength=0.1)
            d1, d2 = abs(float(loss_fwd) - float(o_fwd)), abs(float(loss_exp) - float(o_exp))
            print(f"  {tag} {obj}: forward loss {float(loss_fwd):.6e} (t = {t_fwd.tolist()}), p_losses {float(loss_exp):.6e}; "
                  f"oracle-vs-reference {d1:.3e} / {d2:.3e}")
            assert d1 <= 1e-6 * max(1.0, abs(float(loss_fwd))) and d2 <= 1e-6 * max(1.0, abs(float(loss_exp)))
            out[f"{tag}_{obj}_t_fwd"], out[f"{tag}_{obj}_loss_fwd"] = t_fwd.numpy(), np.float32(float(loss_fwd))
            out[f"{tag}_{obj}_per_fwd"] = to_np(per_fwd)
            out[f"{tag}_{obj}_t_exp"], out[f"{tag}_{obj}_loss_exp"] = t_exp.numpy(), np.float32(float(loss_exp))
            out[f"{tag}_{obj}_per_exp"] = to_np(per_exp)
    save("g15_p_losses", **out)


def g0_inventory(ddpm):
    print("G0 parameter inventory")
    lines = []
    for tag, cfg in [("mnist", CFG_MNIST), ("mri", CFG_MRI), ("mvtec", CFG_MVTEC)]:
        sd = sd_torch(cfg)
        build_reference_unet(ddpm, cfg, sd)
        n = sum(v.numel() for v in sd.values())
        lines.append(f"{tag} {len(sd)} {n}")
        print("  ", lines[-1])
    with open(os.path.join(GOLD, "g0_inventory.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")


def g17(ddpm):
    """The reference's remaining Unet constructor options (ddpm.py:294-300), which no shipped caller sets: learned_variance
    (out_dim doubles, :394), learned_sinusoidal_cond / random_fourier_features (RandomOrLearnedSinusoidalPosEmb, :151-165) and
    self_condition (accepted by the constructor; the forward then fails in init_conv, :406-413).  Forward outputs of the real
    reference with the first two on, on the MNIST-shaped net; the oracle must reproduce them."""
    print("G17 Unet constructor options: learned_variance + learned Fourier time features; self_condition")
    cfg = weights.UnetConfig(dim_mults=(1, 2, 4), full_attn=(False, False, True), mode="mnist", out_dim=2, learned_sinusoidal_dim=16)
    sd = sd_torch(cfg)
    out = {}
    for tag, kw in (("learned", dict(learned_sinusoidal_cond=True)), ("random", dict(random_fourier_features=True))):
        ref = ddpm.Unet(dim=cfg.dim, init_dim=cfg.init_dim, dim_mults=cfg.dim_mults, channels=cfg.channels, full_attn=cfg.full_attn,
                        mode=cfg.mode, learned_variance=True, learned_sinusoidal_dim=16, **kw)
        assert ref.out_dim == 2 and ref.random_or_learned_sinusoidal_cond
        ref_sd = ref.state_dict()
        assert list(ref_sd.keys()) == list(sd.keys()), set(ref_sd) ^ set(sd)
        for k in ref_sd:
            assert tuple(ref_sd[k].shape) == tuple(sd[k].shape), k
        ref.load_state_dict(sd)
        ref.eval()
        B, H = 2, 28
        x = torch.from_numpy(rng.randn((B, 1, H, H), 17, 100))
        cond = torch.from_numpy(rng.uniform((B, 1, H, H), 17, 101, 0.0, 2.0))
        for t in (0, 7, 99):
            tv = torch.full((B,), t, dtype=torch.long)
            with torch.no_grad():
                y_ref = ref(x, cond, tv)
                y_orc = unet_ref.unet_forward(sd, cfg, x, cond, tv)
            d = maxdiff(y_ref, y_orc)
            print(f"  {tag} t={t}: out {tuple(y_ref.shape)}, oracle-vs-reference {d:.2e}")
            assert tuple(y_ref.shape) == (B, 2, H, H) and d <= 1e-5, (tag, t, d)
            if tag == "learned":
                out[f"t{t}_out"] = y_ref.numpy()
        try:                          # the reference's own GaussianDiffusion refuses both (ddpm.py:515-516)
            _ref_diffusion(ddpm, base_config(data="mnist"), ref, H, 10, "sigmoid", "pred_x0", None)
            raise SystemExit("GaussianDiffusion accepted a learned-variance / learned-sinusoidal Unet")
        except AssertionError:
            pass
    sc = ddpm.Unet(dim=32, init_dim=32, dim_mults=cfg.dim_mults, channels=1, full_attn=cfg.full_attn, mode="mnist", self_condition=True)
    try:
        with torch.no_grad():
            sc(x, cond, torch.zeros(B, dtype=torch.long))
        raised = ""
    except RuntimeError as e:
        raised = str(e)
    print("  self_condition=True forward:", raised[:120] or "no error")
    assert "channels" in raised
    out["selfcond_forward_raises"] = np.array([1])
    out["shape"] = np.array([B, 1, H, 1])
    save("g17_unet_options", **out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--skip-long", action="store_true")
    a = ap.parse_args()
    torch.set_num_threads(os.cpu_count())
    ddpm = import_reference()
    os.makedirs(GOLD, exist_ok=True)
    todo = [("G0", g0_inventory), ("G1", g1), ("G2", g2), ("G3", g3), ("G4", g4), ("G6", g6),
            ("G7", g7), ("G8", g8), ("G9", g9), ("G10", g10), ("G12", g12), ("G14", g14), ("G15", g15), ("G17", g17), ("G5", g5), ("G11", g11), ("G13", g13), ("G16", g16)]
    only = set(filter(None, a.only.split(",")))
    for name, fn in todo:
        if only and name not in only:
            continue
        if a.skip_long and name in ("G5", "G11", "G13", "G16"):
            continue
        t0 = time.time()
        fn(ddpm)
        print(f"  [{name} done in {time.time()-t0:.1f}s]")


if __name__ == "__main__":
    main()
