"""The sharded drivers on a real GPU: world size 1 through RCCL (torch.distributed "nccl"), and the property that
makes N > 1 correct by construction -- a shard run with ``noise_offset`` reproduces the unsharded batch's samples."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu

import localdiffusion_hallucination_amd as ldh                   # noqa: E402
from localdiffusion_hallucination_amd import dist as ldist, rng  # noqa: E402
from test_hip_sampler import make                                 # noqa: E402


@pytest.fixture(scope="module")
def world1():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield
    dist.destroy_process_group()


@pytest.mark.parametrize("noise", ["device", "host"])
def test_a_shard_with_noise_offset_reproduces_the_unsharded_batch(noise):
    """Samples [lo, hi) of a batch, run alone with noise_offset = lo*C*H*W, are bit-identical to the same samples inside
    the whole batch when the kernels pick the same tile variants (they do for 2 vs 4 samples of 32x32): every
    per-sample quantity -- x_T, z_t, GroupNorm statistics, attention -- is computed per batch element."""
    B, H, T = 4, 32, 12
    cond = torch.from_numpy(rng.uniform((B, 1, H, H), 41, 1, 0.0, 2.0)).cuda()
    mask = torch.zeros(B, 1, H, H)
    mask[:, :, :, :H // 4] = 1.0
    kw = dict(data="mri", branch_out=True, start_intermediate=True, start_timestep=3, mask_x=True, ood_AD=True)
    gd = make(dict(mode="mri"), kw, H, T)
    gd.noise_source, gd.sub_batches = noise, 1
    whole = gd.sample(cond, None, batch_size=B, mask=mask.cuda(), min_max_val=(0.0, 2.0)).cpu().numpy()
    parts = []
    for lo, hi in ((0, 2), (2, 4)):
        gd.noise_offset = lo * 1 * H * H
        parts.append(gd.sample(cond[lo:hi], None, batch_size=hi - lo, mask=mask[lo:hi].cuda(), min_max_val=(0.0, 2.0)).cpu().numpy())
    gd.noise_offset = 0
    d = float(np.abs(np.concatenate(parts) - whole).max())
    print(f"shards with noise_offset vs the whole batch ({noise} noise): max-abs {d:.3e}")
    assert d <= 1e-5


def test_sharded_drivers_world1(world1):
    B, H, T, K = 2, 32, 10, 4
    # (a) image-sharded branch -> fusion -> joint == the plain call
    cond = torch.from_numpy(rng.uniform((B, 1, H, H), 42, 1, 0.0, 2.0)).cuda()
    mask = torch.zeros(B, 1, H, H)
    mask[:, :, :, :H // 4] = 1.0
    kw = dict(data="mri", branch_out=True, start_intermediate=True, start_timestep=3, mask_x=True, ood_AD=True)
    gd = make(dict(mode="mri"), kw, H, T)
    gd.noise_source = "device"
    want = gd.sample(cond, None, batch_size=B, mask=mask.cuda(), min_max_val=(0.0, 2.0))
    got = ldist.sample_images_sharded(gd, cond, None, mask.cuda(), (0.0, 2.0))
    assert got.shape == want.shape and torch.equal(got, want)
    gd2 = make(dict(mode="mri"), dict(kw, start_intermediate=False), H, T)
    gd2.noise_source = "device"
    want = gd2.sample(cond, None, batch_size=B, mask=mask.cuda(), min_max_val=(0.0, 2.0))
    got = ldist.sample_images_sharded(gd2, cond, None, mask.cuda(), (0.0, 2.0))
    assert tuple(want.shape) == (2, B, 1, H, H) and torch.equal(got, want)
    # (b) independent patches + one all-gather + ld_recompose == sum_k x_k * m_k
    gd3 = make(dict(mode="mri"), dict(data="mri"), H, T)
    gd3.noise_source = "device"
    masks = torch.zeros(K, 1, H, H)
    for k in range(K):
        masks[k, :, :, k * (H // K):(k + 1) * (H // K)] = 1.0
    conds = torch.from_numpy(rng.uniform((B * K, 1, H, H), 43, 1, 0.0, 2.0)).cuda()
    img = ldist.sample_patches_sharded(gd3, conds, (0.0, 2.0), B, K, masks)
    x = gd3.sample(conds, None, batch_size=B * K, min_max_val=(0.0, 2.0)).reshape(B, K, 1, H, H)
    ref = (x * masks.cuda()[None]).sum(1)
    assert tuple(img.shape) == (B, 1, H, H) and float((img - ref).abs().max()) <= 1e-6


def test_cfg4_shaped_patch_sharding_world1(world1):
    """cfg4's unit of work through the sharded driver at world size 1: 8 images x 8 band masks of 3x256x256 in bf16,
    ONE all-gather (RCCL) in the storage dtype, ld_recompose == sum_k x_k m_k of the plain batch."""
    B, K, H, T = 8, 8, 256, 5
    net = ldh.Unet(dim=32, init_dim=32, channels=3, out_dim=3, mode="mvtec", compute_dtype="bf16")
    from localdiffusion_hallucination_amd import weights
    net.load_state_dict({k: torch.from_numpy(v) for k, v in weights.procedural_state_dict(net.cfg, 0).items()})
    cfg = dict(branch_out=False, start_intermediate=False, start_timestep=2, data="mvtec", mask_x=False, mask_cond=False,
               ood_AD=False, ood_confidence=False, classifier=False, use_gt=False, use_gt_timestep=100)
    gd = ldh.GaussianDiffusion(cfg, net, image_size=H, timesteps=T, beta_schedule="sigmoid", objective="pred_x0",
                               auto_normalize=False).to("cuda")
    gd.noise_source = "device"
    masks = torch.zeros(K, 1, H, H)
    for k in range(K):
        masks[k, :, :, k * (H // K):(k + 1) * (H // K)] = 1.0
    imgs = torch.from_numpy(rng.uniform((B, 3, H, H), 65, 1, 0.0, 2.0))
    conds = torch.stack([imgs[i] * (masks[k] if k == 0 else torch.clip(masks[k], 0.95, 1.0))
                         for i in range(B) for k in range(K)]).cuda()
    x = gd.sample(conds, None, batch_size=B * K, min_max_val=(0.0, 2.0)).reshape(B, K, 3, H, H)
    ref = (x * masks.cuda()[None]).sum(1)
    img = ldist.sample_patches_sharded(gd, conds, (0.0, 2.0), B, K, masks)
    assert tuple(img.shape) == (B, 3, H, H) and float((img - ref).abs().max()) <= 1e-6
    # the storage dtype on the wire: the samples are rounded to bf16 once, after the chain
    img16 = ldist.sample_patches_sharded(gd, conds, (0.0, 2.0), B, K, masks, gather_dtype=torch.bfloat16)
    ref16 = (x.to(torch.bfloat16).float() * masks.cuda()[None]).sum(1)
    assert float((img16 - ref16).abs().max()) <= 1e-6 and float((img16 - ref).abs().max()) <= 2.0 * 2 ** -8


def test_ld_allgather_world1_through_the_c_abi():
    """The C ABI's own RCCL communicator (dlopen'ed librccl: ld_comm_unique_id / ld_comm_init / ld_allgather) at world
    size 1 on the GPU, alone and as the collective of gather_patches; N > 1 is the same call with the id shared."""
    comm = ldist.LdComm.bootstrap(world=1, rank=0)
    x = torch.from_numpy(rng.uniform((5, 3, 16, 16), 77, 1, 0.0, 2.0)).cuda()
    y = torch.empty_like(x)
    comm.all_gather(x, y)
    torch.cuda.synchronize()
    assert torch.equal(x, y)
    assert torch.equal(ldist.gather_patches(x, 5, comm=comm), x)
    comm.close()


def _emulate_ranks(gd, cond, masks, world):
    """dist.sample_kmask_sharded's arithmetic with the ranks run one after the other in this process (the collectives
    are concatenations): units over `world` ranks, one payload, images over `world` ranks."""
    B, K = masks.shape[:2]
    parts, where = [], None
    for r in range(world):
        lo, hi = ldist.shard_bounds(K * B, world, r)
        p, w = gd.kmask_branch_units(cond, masks, (0.0, 2.0), lo, hi)
        where = w if where is None else where
        assert w == where
        parts.append(p)
    pay = torch.cat(parts, 0)
    _, fuse, _ = gd.kmask_flags(masks)
    if not fuse:
        out = pay[:, 0].reshape(K, B, *pay.shape[2:])
    else:
        out = torch.cat([gd.kmask_fuse_joint(cond, masks, (0.0, 2.0), pay, where, *ldist.shard_bounds(B, world, r)) for r in range(world)], 0)
    gd.advance_call_state(masks)
    return out.cpu().numpy()


@pytest.mark.parametrize("ddim", [False, True])
def test_kmask_units_sharded_equal_the_unsharded_loop_and_the_reference_goldens(golden, ddim):
    """SURVEY 8e, mid-chain fusion across ranks: the K branch-patches of every image spread over 1, 2 and 3 emulated
    ranks, ONE exchange of (x_t, x0_hat) at t = start_timestep, the joint steps by image.  K = 2 with masks
    [m, 1 - (m >= 1)] reproduces the reference's own goldens (G6 DDPM / G7 DDIM); K = 4 equals the unsharded K-mask loop
    (fp32; a shard's plan has another batch size, so tile variants -- not values per sample -- may differ: <= 1e-5)."""
    if not ddim:
        g = golden("g6_branch_fusion")
        H, B, T, S = 32, 2, 50, None
        cond = torch.from_numpy(rng.uniform((B, 1, H, H), 6, 1, 0.0, 2.0)).cuda()
        mask = torch.zeros(B, 1, H, H)
        mask[:, :, :, :H // 4] = 1.0
        ref, kw = g[f"mri{H}_final"], dict(mode="mri")
    else:
        g = golden("g7_ddim")
        mask = torch.from_numpy(g["mask"])
        H, B, T, S = mask.shape[-1], 1, 1000, 50
        cond = torch.from_numpy(rng.uniform((B, 1, H, H), 7, 1, 0.0, 2.0)).cuda()
        ref, kw = g["fused_final"], dict(mode="mri")
    masks2 = torch.cat([mask, 1.0 - (mask >= 1.0).float()], 1).cuda()
    conf = dict(data="mri", branch_out=True, start_intermediate=True, start_timestep=2, mask_x=True)
    for world in (1, 2, 3):
        gd = make(kw, conf, H, T, S)
        got = _emulate_ranks(gd, cond, masks2, world)
        d = float(np.abs(got - ref).max())
        print(f"K=2 over {world} emulated ranks ({'DDIM' if ddim else 'DDPM'}) vs the reference golden: max-abs {d:.3e}")
        assert d <= 1e-3
    # K = 4, fused and kept apart, against the unsharded loop of the same object
    K, B, H = 4, 2, 32
    masks = torch.zeros(B, K, H, H)
    for k in range(K):
        masks[:, k, :, k * (H // K):(k + 1) * (H // K)] = 1.0
    masks = masks.cuda()
    cond = torch.from_numpy(rng.uniform((B, 1, H, H), 44, 1, 0.0, 2.0)).cuda()
    for fuse in (True, False):
        conf = dict(data="mri", branch_out=True, start_intermediate=fuse, start_timestep=3, mask_x=True)
        gd = make(dict(mode="mri"), conf, H, 50 if ddim else 14, 10 if ddim else None)
        whole = gd.sample(cond, None, batch_size=B, mask=masks, min_max_val=(0.0, 2.0))
        whole = (torch.stack(whole) if isinstance(whole, list) else whole).cpu().numpy()
        for world in (2, 3, 8):
            gd.reset_call_state()
            got = _emulate_ranks(gd, cond, masks, world)
            assert got.shape == whole.shape
            d = float(np.abs(got - whole).max())
            print(f"K=4 fuse={fuse} over {world} emulated ranks vs the unsharded loop: max-abs {d:.3e}")
            assert d <= 1e-5


def test_kmask_units_sharded_with_the_use_gt_start():
    """The shortened chain of ddpm.py:937-944 (x_T = q_sample(hr, use_gt_timestep), start at use_gt_timestep - 1) through the
    two halves: units over 3 emulated ranks against the unsharded K-mask loop, for the data mode whose OOD branch skips the
    denoiser as well (its state still steps with the shared draw)."""
    K, B, H, T = 3, 2, 32, 40
    masks = torch.zeros(B, K, H, H)
    masks[:, 0, :, :8] = 1.0
    masks[:, 1, :, 8:20] = 1.0
    masks[:, 2, :, 20:] = 1.0
    masks = masks.cuda()
    cond = torch.from_numpy(rng.uniform((B, 1, H, H), 47, 1, 0.0, 2.0)).cuda()
    hr = torch.from_numpy(rng.uniform((B, 1, H, H), 48, 1, 0.0, 2.0)).cuda()
    for data in ("mri", "mnist"):
        conf = dict(data=data, branch_out=True, start_intermediate=True, start_timestep=3, mask_x=True, use_gt=True, use_gt_timestep=12)
        gd = make(dict(mode="mri"), conf, H, T)
        want = gd.sample(cond, hr, batch_size=B, mask=masks, min_max_val=(0.0, 2.0)).cpu().numpy()
        gd.reset_call_state()
        parts, where = [], None
        for r in range(3):
            lo, hi = ldist.shard_bounds(K * B, 3, r)
            p, where = gd.kmask_branch_units(cond, masks, (0.0, 2.0), lo, hi, gt=hr)
            parts.append(p)
        assert where[0] == 3                                   # the exchange sits at t = start_timestep
        pay = torch.cat(parts, 0)
        got = torch.cat([gd.kmask_fuse_joint(cond, masks, (0.0, 2.0), pay, where, *ldist.shard_bounds(B, 3, r)) for r in range(3)], 0)
        d = float(np.abs(got.cpu().numpy() - want).max())
        print(f"use_gt start, {data}: units over 3 emulated ranks vs the unsharded loop: max-abs {d:.3e}")
        assert d <= 1e-5


def test_kmask_sharded_world1_through_ld_allgather():
    """dist.sample_kmask_sharded end to end at world size 1 with the C ABI's own communicator (ld_comm_* / ld_allgather,
    RCCL through dlopen): both exchanges go through the collective; equal to the plain call, for the non-MRI data mode as
    well (the OOD branch's prediction is its conditioning, ddpm.py:704-708: those units skip the denoiser), in the storage
    dtype on the wire too."""
    K, B, H, T = 3, 2, 32, 12
    masks = torch.zeros(B, K, H, H)
    masks[:, 0, :, :8] = 1.0
    masks[:, 1, :, 8:20] = 1.0
    masks[:, 2, :, 20:] = 1.0
    masks = masks.cuda()
    cond = torch.from_numpy(rng.uniform((B, 1, H, H), 46, 1, 0.0, 2.0)).cuda()
    with ldist.LdComm.bootstrap(world=1, rank=0) as comm:
        for data in ("mri", "mvtecGray"):
            conf = dict(data=data, branch_out=True, start_intermediate=True, start_timestep=3, mask_x=True, ood_AD=True)
            gd = make(dict(mode="mri"), conf, H, T)
            gd.noise_source = "device"
            want = gd.sample(cond, None, batch_size=B, mask=masks, min_max_val=(0.0, 2.0))
            got = ldist.sample_kmask_sharded(gd, cond, None, masks, (0.0, 2.0), comm=comm)
            assert got.shape == want.shape
            d = float((got - want).abs().max())
            print(f"sample_kmask_sharded ({data}) through ld_allgather vs sample(): max-abs {d:.3e}")
            assert d <= 1e-5
            got16 = ldist.sample_kmask_sharded(gd, cond, None, masks, (0.0, 2.0), comm=comm, gather_dtype=torch.float16)
            assert float((got16 - want).abs().max()) <= 2e-2          # fp16 on the wire at the exchange, three joint steps behind it
