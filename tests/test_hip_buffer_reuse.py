"""Tuning.buffer_reuse: a sampler plan's activations placed in ONE pool by liveness (unet._Plan._pool_buffers) must change
nothing but addresses: the same samples bit for bit, in every storage type, through the branch / fusion phases, DDIM, the
sub-batch runner with replayed graphs and the K-mask loop -- and the pool must be a fraction of the per-layer buffers."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import localdiffusion_hallucination_amd as ldh                   # noqa: E402
from localdiffusion_hallucination_amd import rng, weights        # noqa: E402
from localdiffusion_hallucination_amd.tuning import Tuning        # noqa: E402

MNIST = dict(dim_mults=(1, 2, 4), full_attn=(False, False, True), mode="mnist")
MRI = dict(mode="mri")


def _gd(kw, config, H, T, dtype, reuse, S=None, sub_batches=2):
    net = ldh.Unet(dim=32, init_dim=32, compute_dtype=dtype, tuning=Tuning.from_env(buffer_reuse=reuse, sub_batches=sub_batches), **kw)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in weights.procedural_state_dict(net.cfg, 0).items()})
    cfg = dict(branch_out=False, start_intermediate=False, start_timestep=2, data="mri", mask_x=False, mask_cond=False,
               ood_AD=False, ood_confidence=False, classifier=False, use_gt=False, use_gt_timestep=100)
    cfg.update(config)
    gd = ldh.GaussianDiffusion(cfg, net, image_size=H, timesteps=T, beta_schedule="sigmoid", objective="pred_x0",
                               auto_normalize=False, sampling_timesteps=S).to("cuda")
    gd.noise_source = "device"
    return gd, net


def _sample(gd, cond, mask, B):
    out = gd.sample(cond.cuda(), None, batch_size=B, mask=None if mask is None else mask.cuda(), min_max_val=(0.0, 2.0))
    if isinstance(out, list):
        out = torch.stack(out)
    return out.cpu().numpy()


def _pools(net):
    return [p.pool_stats for p in net._plans.values() if getattr(p, "pool_stats", None)]


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16"])
def test_plain_chain_is_bit_equal_and_the_pool_is_small(dtype):
    cond = torch.from_numpy(rng.uniform((4, 1, 64, 64), 31, 1, 0.0, 2.0))
    outs = {}
    for reuse in (False, True):
        gd, net = _gd(MRI, dict(data="mri"), 64, 12, dtype, reuse)
        outs[reuse] = _sample(gd, cond, None, 4)
        if reuse:
            pools = _pools(net)
            assert pools, "no plan was pooled"
            for ps in pools:
                print(dtype, ps)
                assert ps["bytes_pool"] < 0.45 * ps["bytes_unshared"], ps
    assert np.isfinite(outs[True]).all() and np.array_equal(outs[False], outs[True])


@pytest.mark.parametrize("ddim", [False, True])
def test_branch_fusion_and_ddim_are_bit_equal(ddim):
    H, T = 64, 10
    cond = torch.from_numpy(rng.uniform((2, 1, H, H), 32, 1, 0.0, 2.0))
    mask = torch.zeros(2, 1, H, H)
    mask[..., 16:40, 20:44] = 1.0
    cfg = dict(data="mri", branch_out=True, start_timestep=3, mask_x=True)
    outs = {}
    for reuse in (False, True):
        gd, _ = _gd(MRI, cfg, H, T, "bf16", reuse, S=6 if ddim else None)
        outs[reuse] = _sample(gd, cond, mask, 2)
    assert np.isfinite(outs[True]).all() and np.array_equal(outs[False], outs[True])


def test_mnist_one_stream_is_bit_equal():
    cond = torch.from_numpy(rng.uniform((3, 1, 28, 28), 33, 1, 0.0, 2.0))
    outs = {}
    for reuse in (False, True):
        gd, _ = _gd(MNIST, dict(data="mnist"), 28, 9, "fp16", reuse, sub_batches=1)
        outs[reuse] = _sample(gd, cond, None, 3)
    assert np.isfinite(outs[True]).all() and np.array_equal(outs[False], outs[True])


def test_kmask_loop_is_bit_equal():
    """The K-mask branch loop (three masks per image, fusion + joint steps) on pooled plans (the docstring claimed it)."""
    H, T, K = 64, 10, 3
    cond = torch.from_numpy(rng.uniform((2, 1, H, H), 34, 1, 0.0, 2.0))
    mask = torch.zeros(2, K, H, H)
    for k in range(K):
        mask[:, k, :, k * 20:(k + 1) * 20 + (4 if k == K - 1 else 0)] = 1.0
    cfg = dict(data="mri", branch_out=True, start_intermediate=True, start_timestep=3, mask_x=True)
    outs = {}
    for reuse in (False, True):
        gd, _ = _gd(MRI, cfg, H, T, "bf16", reuse)
        outs[reuse] = _sample(gd, cond, mask, 2)
    assert np.isfinite(outs[True]).all() and np.array_equal(outs[False], outs[True])


def _verify_pair(kw, config, H, T, dtype, cond, mask, B, S=None, sub_batches=2):
    outs, stats = {}, None
    for verify in (False, True):
        net = ldh.Unet(dim=32, init_dim=32, compute_dtype=dtype,
                       tuning=Tuning.from_env(buffer_reuse=True, pool_verify=verify, sub_batches=sub_batches), **kw)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in weights.procedural_state_dict(net.cfg, 0).items()})
        cfg = dict(branch_out=False, start_intermediate=False, start_timestep=2, data="mri", mask_x=False, mask_cond=False,
                   ood_AD=False, ood_confidence=False, classifier=False, use_gt=False, use_gt_timestep=100)
        cfg.update(config)
        gd = ldh.GaussianDiffusion(cfg, net, image_size=H, timesteps=T, beta_schedule="sigmoid", objective="pred_x0",
                                   auto_normalize=False, sampling_timesteps=S).to("cuda")
        gd.noise_source = "device"
        outs[verify] = _sample(gd, cond, mask, B)
        if verify:
            stats = _pools(net)
    assert stats and all(ps.get("verify_poisoned_buffers", 0) > 50 for ps in stats), stats
    assert np.isfinite(outs[True]).all(), "a launch read a buffer behind its last DECLARED use (it was poisoned with NaNs there)"
    assert np.array_equal(outs[False], outs[True])
    return stats


def test_pool_verify_cfg3_shape():
    """LD_POOL_VERIFY: every pooled buffer is overwritten with NaNs right behind the launch of its last declared use.  At
    cfg3's shape (8 patches of 3x256x256, bf16, two replayed sub-batches of 4) the samples stay finite and bit-equal, and the
    pool is still the 100.7 MB peak live set of a 4-patch plan."""
    cond = torch.from_numpy(rng.uniform((8, 3, 256, 256), 35, 1, 0.0, 2.0))
    stats = _verify_pair(dict(channels=3, out_dim=3, mode="mvtec"), dict(data="mvtec"), 256, 6, "bf16", cond, None, 8)
    four = [ps for ps in stats if 50e6 < ps["bytes_pool"] < 150e6]          # the two 4-patch sub-batch plans (the 8-patch parent plan: 201.3 MB)
    print(stats)
    assert four and all(abs(ps["bytes_pool"] - 100.7e6) < 0.6e6 and ps["bytes_pool"] == ps["bytes_peak_live"] for ps in four), stats


def test_pool_verify_cfg5_shape():
    """... cfg5's shape (1x512x512, fp16, DDIM, OOD + IND branches with fusion, full attention over 4,096 tokens)."""
    H = 512
    cond = torch.from_numpy(rng.uniform((1, 1, H, H), 36, 1, 0.0, 2.0))
    yy, xx = np.mgrid[0:H, 0:H]
    mask = torch.from_numpy((((yy - H / 2) ** 2 + (xx - H / 2) ** 2) <= 64 ** 2).astype(np.float32))[None, None]
    _verify_pair(MRI, dict(data="mri", branch_out=True, start_intermediate=True, start_timestep=2, mask_x=True, ood_AD=True),
                 H, 1000, "fp16", cond, mask, 1, S=8)


def test_pool_verify_64_patch_shape():
    """... cfg4's per-GPU share (64 patches of 3x256x256, bf16, two sub-batches of 32)."""
    cond = torch.from_numpy(rng.uniform((64, 3, 256, 256), 37, 1, 0.0, 2.0))
    _verify_pair(dict(channels=3, out_dim=3, mode="mvtec"), dict(data="mvtec"), 256, 5, "bf16", cond, None, 64)


def test_an_undeclared_pointer_fails_loudly(monkeypatch):
    """A launch that touches a pooled buffer without declaring it (here: a raw launch closing over the first block's output)
    must stop the plan build -- its liveness would be wrong and the buffer would be re-used while the launch still reads it.
    And a declared buffer held only as a plain integer address cannot be re-pointed at the pool: also an error."""
    from localdiffusion_hallucination_amd import unet as unet_mod
    orig = unet_mod._Plan._build_main

    def make(kind):
        def patched(self):
            orig(self)
            victim = self._track[3]
            if kind == "undeclared":
                self._raw(self.ops_main, lambda st, victim=victim: None, "rogue launch")
            else:
                addr = victim.data_ptr()
                self._raw(self.ops_main, lambda st, addr=addr: None, "integer launch", reads=[victim])
        return patched
    for kind, pat in (("undeclared", "rogue launch.*does not\\s+declare"), ("integer", "integer launch.*no patchable reference")):
        monkeypatch.setattr(unet_mod._Plan, "_build_main", make(kind))
        net = ldh.Unet(dim=32, init_dim=32, compute_dtype="bf16", tuning=Tuning.from_env(buffer_reuse=True), **MRI)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in weights.procedural_state_dict(net.cfg, 0).items()})
        net = net.to("cuda")
        with pytest.raises(RuntimeError, match=pat):
            net.plan(1, 64, 64, table_T=10)
    monkeypatch.setattr(unet_mod._Plan, "_build_main", orig)
