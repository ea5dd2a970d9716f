"""Multi-GPU execution of the local-diffusion path: independent local patches are sharded across
ranks, every rank runs the reverse loop on its shard with NO traffic inside the T-loop, and ONE
all-gather (RCCL over xGMI; ``torch.distributed`` backend "nccl" is RCCL on ROCm) brings the
patches of every image together for mask recomposition.

The reference has no multi-GPU sampling path at all (its only collectives are HF-Accelerate DDP
training calls, /root/reference/ddpm.py:1462,1553,1557); the unit being sharded here is the
"branch" tensor of ``model_predictions`` (/root/reference/ddpm.py:693-695) generalised to K masks.

The sharding arithmetic and the gather are backend-agnostic (they are exercised with gloo on CPU
in tests/test_dist.py); recomposition on a GPU uses the HIP kernel ``ld_recompose``.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_items, world, rank):
    """Contiguous block partition: rank r owns items [lo, hi); sizes differ by at most one."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def patch_owner(p, n_items, world):
    """Inverse of shard_bounds: the rank that owns patch index p."""
    base, rem = divmod(n_items, world)
    split = rem * (base + 1)
    if p < split:
        return p // (base + 1)
    return rem + (p - split) // max(base, 1)


def shard_patches(x, world=None, rank=None):
    """x: [P, ...] (all patches, identical on every rank) -> this rank's contiguous shard."""
    world = dist.get_world_size() if world is None else world
    rank = dist.get_rank() if rank is None else rank
    lo, hi = shard_bounds(x.shape[0], world, rank)
    return x[lo:hi]


def gather_patches(local, n_items, group=None):
    """All-gather ragged shards back into [n_items, ...] on every rank (one collective).

    Shards are padded to the largest shard so a single ``all_gather_into_tensor`` suffices
    (payload: <= 201 MB at 512 patches of 3x256x256 bf16 -- latency-, not bandwidth-bound on xGMI).
    """
    world = dist.get_world_size(group)
    sizes = [shard_bounds(n_items, world, r) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    out = torch.empty((world * mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad.contiguous(), group=group)
    parts = [out[r * mx: r * mx + (hi - lo)] for r, (lo, hi) in enumerate(sizes)]
    return torch.cat(parts, 0)


def recompose(patches, masks):
    """patches [B, K, C, H, W] fp32, masks [K, 1, H, W] -> [B, C, H, W] = sum_k patches[:,k] * (mask_k >= 1).

    On a GPU this is the ``ld_recompose`` HIP kernel; there is no CPU implementation on the product
    path (tests compare against the oracle's formula)."""
    if patches.device.type != "cuda":
        raise RuntimeError("recompose runs on the GPU (ld_recompose); there is no CPU fallback")
    from . import _cabi as cabi
    B, K, C, H, W = patches.shape
    out = torch.empty(B, C, H, W, dtype=torch.float32, device=patches.device)
    p = patches.to(torch.float32).contiguous()
    m = masks.reshape(K, H * W).to(patches.device, torch.float32).contiguous()
    cabi.check(cabi.lib().ld_recompose(p.data_ptr(), m.data_ptr(), out.data_ptr(), B, K, C, H * W,
                                       torch.cuda.current_stream().cuda_stream), "recompose")
    return out


def _sample_shard(diffusion, lo, hi, per_sample_elems, call):
    """Run ``call()`` with the diffusion object's noise stream positioned at sample ``lo`` of the global batch, so
    that the shard draws exactly the x_T and z_t values the unsharded batch would draw for samples [lo, hi): the
    sharded result equals the single-GPU result sample for sample."""
    keep = getattr(diffusion, "noise_offset", 0)
    diffusion.noise_offset = keep + lo * per_sample_elems
    try:
        return call()
    finally:
        diffusion.noise_offset = keep


def sample_patches_sharded(diffusion, conds, min_max_val, n_images, k_masks, masks):
    """Independent local patches: run ``diffusion.sample`` on this rank's contiguous shard
    of the [n_images*k_masks] patch list with no traffic inside the T-loop, ONE all-gather, recomposition by the
    masks.  conds: [n_images*k_masks, Cc, H, W] (already masked per patch), identical on all ranks.
    Returns the recomposed images [n_images, C, H, W] on every rank."""
    P = n_images * k_masks
    world, rank = dist.get_world_size(), dist.get_rank()
    lo, hi = shard_bounds(P, world, rank)
    C, H = diffusion.channels, diffusion.image_size
    if hi > lo:
        x = _sample_shard(diffusion, lo, hi, C * H * H, lambda: diffusion.sample(
            conds[lo:hi], None, batch_size=hi - lo, mask=None, min_max_val=min_max_val))
    else:                                                    # more ranks than patches: replicas idle
        x = conds.new_zeros((0, C, H, H), dtype=torch.float32)
    allx = gather_patches(x.to(torch.float32), P)
    return recompose(allx.reshape(n_images, k_masks, *allx.shape[1:]), masks)


def sample_images_sharded(diffusion, cond_img, gt, masks, min_max_val, **sample_kw):
    """The reference's own branch -> fusion -> joint path (ddpm.py:779-810, 955-970; cfg5) across ranks: the unit is
    the IMAGE, so the OOD and the IND branch of an image stay on one rank through the fusion step and the joint
    steps, and no collective is needed until the samples are complete.  Rank r runs
    ``diffusion.sample(cond_img[lo:hi], gt[lo:hi], mask=masks[lo:hi])`` on its contiguous block of images with the
    noise stream positioned at image ``lo``, then ONE all-gather returns the full batch on every rank:
    [n, C, H, W] (fused), [2, n, C, H, W] (branches kept apart, ddpm.py:965-970) or a list of two [n, C, H, W]
    tensors (DDIM without fusion, :1069-1075) -- whatever ``sample`` returns for the unsharded batch."""
    n = cond_img.shape[0]
    world, rank = dist.get_world_size(), dist.get_rank()
    lo, hi = shard_bounds(n, world, rank)
    C, H = diffusion.channels, diffusion.image_size
    out = None
    if hi > lo:
        out = _sample_shard(diffusion, lo, hi, C * H * H, lambda: diffusion.sample(
            cond_img[lo:hi], None if gt is None else gt[lo:hi], batch_size=hi - lo,
            mask=None if masks is None else masks[lo:hi], min_max_val=min_max_val, **sample_kw))
    # the layout of the result is a function of the flags, not of the shard: every rank (also an idle one) derives it
    branch, fuse, _ = diffusion._flags(masks)
    as_list = diffusion.is_ddim_sampling and branch and not fuse
    stacked = (not diffusion.is_ddim_sampling) and (not fuse) and diffusion.branch_out
    dev = cond_img.device
    if out is None:
        out = torch.zeros((2, 0, C, H, H) if (as_list or stacked) else (0, C, H, H), dtype=torch.float32, device=dev)
    elif isinstance(out, (list, tuple)):
        out = torch.stack(list(out), 0)
    out = out.to(torch.float32)
    if out.dim() == 5:                                       # [2, n_loc, ...]: gather along the image axis
        full = gather_patches(out.transpose(0, 1).contiguous(), n).transpose(0, 1).contiguous()
        return [full[0], full[1]] if as_list else full
    return gather_patches(out, n)
