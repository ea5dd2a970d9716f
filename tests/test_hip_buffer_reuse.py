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
