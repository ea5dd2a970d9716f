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


class LdComm:
    """The C ABI's own RCCL communicator (``ld_comm_* / ld_allgather``, include/localdiff_hip.h): what a caller without
    a torch process group uses, and an opt-in for ``gather_patches(..., comm=...)``.  The 128-byte unique id made on rank
    0 has to reach every rank through SOME channel; ``bootstrap`` uses the torch process group when there is one (a
    byte-tensor broadcast) or a file path otherwise."""

    def __init__(self, world, rank, unique_id, timeout_s=120.0):
        """``timeout_s`` > 0: the communicator comes up non-blocking and is polled against the deadline
        (``ld_comm_init_timeout``); a peer that never arrives or a stale unique id raises ``TimeoutError`` (the half-built
        communicator is aborted) instead of hanging in ``ncclCommInitRank``.  ``timeout_s`` = 0: the blocking call."""
        import ctypes as C
        from . import _cabi as cabi
        self.world, self.rank = world, rank
        self._comm = C.c_void_p()
        buf = (C.c_char * 128).from_buffer_copy(bytes(unique_id))
        cabi.check(cabi.lib().ld_comm_init_timeout(C.byref(self._comm), buf, world, rank, float(timeout_s or 0.0)), "comm_init")

    @staticmethod
    def make_unique_id():
        import ctypes as C
        from . import _cabi as cabi
        buf = (C.c_char * 128)()
        cabi.check(cabi.lib().ld_comm_unique_id(buf), "comm_unique_id")
        return bytes(buf)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @classmethod
    def bootstrap(cls, world=None, rank=None, id_file=None, run_id=None, timeout_s=120.0):
        """Collective: every rank calls it.  With a torch process group the id travels by broadcast; otherwise rank 0
        writes ``id_file`` and the others poll for it.  ``run_id`` (any string all ranks of THIS run agree on and no
        other run shares: a job id, ``TORCHELASTIC_RUN_ID``) becomes part of the file name; without one the name falls
        back to ``MASTER_PORT``, which torchrun reuses from run to run -- so rank 0 removes a pre-existing file before it
        writes its own and removes its own again as soon as the communicator is up (the init is collective: when it
        returns on rank 0 every rank has read the id); a file can only be left behind by a run that died inside the
        rendezvous.  ``timeout_s`` bounds BOTH waits (round 6): the polling for the file and the communicator's own
        bring-up (``ld_comm_init_timeout``: non-blocking init polled against the deadline, aborted on expiry), so a
        non-zero rank that read such a stale id gets ``TimeoutError`` instead of hanging.  The per-process sequence number
        that keeps successive bootstraps apart advances only when a bootstrap SUCCEEDS: a rank that timed out and retries
        uses the same file name as rank 0 (which never times out on its own file)."""
        import os
        import time
        if dist.is_available() and dist.is_initialized():
            world, rank = dist.get_world_size(), dist.get_rank()
            dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
            t = torch.zeros(128, dtype=torch.uint8, device=dev)
            if rank == 0:
                t.copy_(torch.frombuffer(bytearray(cls.make_unique_id()), dtype=torch.uint8))
            dist.broadcast(t, 0)
            return cls(world, rank, bytes(t.cpu().numpy().tobytes()), timeout_s)
        assert world is not None and rank is not None and (id_file is not None or world == 1)
        if world == 1:
            return cls(1, 0, cls.make_unique_id(), timeout_s)
        if run_id is None:
            from .tuning import rendezvous_run_id
            run_id = rendezvous_run_id()
        if run_id is not None:
            id_file = f"{id_file}.{run_id}"
        # a sequence number per bootstrap of this process (identical on all ranks: bootstrap is collective): a fast rank
        # entering the NEXT bootstrap can no longer find the previous one's file before rank 0 has unlinked it
        seq = getattr(cls, "_bootstraps", 0) + 1           # advanced below, once the communicator is up
        id_file = f"{id_file}.{seq}"
        if rank == 0:
            try:
                os.unlink(id_file)                         # a file of this name can only be a crashed run's: never hand its id out
            except OSError:
                pass
            with open(id_file + ".tmp", "wb") as f:
                f.write(cls.make_unique_id())
            os.replace(id_file + ".tmp", id_file)
        t0 = time.monotonic()
        while not os.path.exists(id_file):
            if time.monotonic() - t0 > timeout_s:
                raise TimeoutError(f"LdComm.bootstrap: rank {rank} saw no {id_file} within {timeout_s} s")
            time.sleep(0.01)
        left = max(timeout_s - (time.monotonic() - t0), 1.0) if timeout_s else 0.0
        comm = cls(world, rank, open(id_file, "rb").read(), left)
        cls._bootstraps = seq
        if rank == 0:
            try:
                os.unlink(id_file)
            except OSError:
                pass
        return comm

    def all_gather(self, send, recv):
        """recv[r*len(send) ...] = rank r's ``send`` (contiguous device tensors), enqueued on the current stream."""
        from . import _cabi as cabi
        assert send.is_contiguous() and recv.is_contiguous() and recv.numel() * recv.element_size() == self.world * send.numel() * send.element_size()
        cabi.check(cabi.lib().ld_allgather(send.data_ptr(), recv.data_ptr(), send.numel() * send.element_size(), self._comm,
                                           torch.cuda.current_stream().cuda_stream), "allgather")

    def close(self):
        from . import _cabi as cabi
        if getattr(self, "_comm", None):
            cabi.check(cabi.lib().ld_comm_destroy(self._comm), "comm_destroy")
            self._comm = None


def _world_rank(comm):
    return (comm.world, comm.rank) if comm is not None else (dist.get_world_size(), dist.get_rank())


def gather_patches(local, n_items, group=None, comm=None, dtype=None):
    """All-gather ragged shards back into [n_items, ...] on every rank (one collective).

    Equal shards go straight into the result (no padding, no copy afterwards); ragged ones are padded to the largest
    shard so that a single ``all_gather_into_tensor`` suffices.  ``dtype``: what travels (the storage
    dtype -- 201 MB of bf16 / fp16 at cfg4's 512 patches of 3x256x256 instead of 402 MB of the fp32 boundary
    tensors); the result comes back in that dtype.  Latency-, not bandwidth-bound on xGMI either way.
    ``comm``: an ``LdComm`` -- the same collective through the C ABI's ``ld_allgather`` instead of torch.distributed.
    """
    if local.device.type == "cuda":
        from . import _cabi as cabi
        with cabi.prof_range("exchange"):
            return _gather_patches(local, n_items, group, comm, dtype)
    return _gather_patches(local, n_items, group, comm, dtype)


def _gather_patches(local, n_items, group, comm, dtype):
    if dtype is not None and local.dtype != dtype:
        local = local.to(dtype)
    world = comm.world if comm is not None else dist.get_world_size(group)
    sizes = [shard_bounds(n_items, world, r) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    equal = all(hi - lo == mx for lo, hi in sizes)
    send = local.contiguous()
    if not equal:
        send = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        send[:local.shape[0]] = local
    out = torch.empty((world * mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    if comm is not None:
        comm.all_gather(send, out)
    else:
        dist.all_gather_into_tensor(out, send, group=group)
    if equal:
        return out
    return torch.cat([out[r * mx: r * mx + (hi - lo)] for r, (lo, hi) in enumerate(sizes)], 0)


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


def _idle_rank_advances(diffusion, masks):
    """A rank without a shard does not call ``sample()``, but the state ``sample()`` carries from one call to the next
    (the reference's mutated ``config['mask_x']``, golden G14) must move on it as on the working ranks: otherwise the next
    sharded call -- with a batch large enough to give this rank work -- would mix first-call and second-call reverse
    processes in one gathered batch."""
    adv = getattr(diffusion, "advance_call_state", None)
    if adv is not None:
        adv(masks)


def sample_patches_sharded(diffusion, conds, min_max_val, n_images, k_masks, masks, gather_dtype=None):
    """Independent local patches: run ``diffusion.sample`` on this rank's contiguous shard
    of the [n_images*k_masks] patch list with no traffic inside the T-loop, ONE all-gather, recomposition by the
    masks.  conds: [n_images*k_masks, Cc, H, W] (already masked per patch), identical on all ranks.
    ``gather_dtype``: the dtype the patches travel in (None: the fp32 boundary tensors; torch.bfloat16 / float16:
    the denoiser's storage dtype, half the payload).  Returns the recomposed images [n_images, C, H, W] on every rank."""
    P = n_images * k_masks
    world, rank = dist.get_world_size(), dist.get_rank()
    lo, hi = shard_bounds(P, world, rank)
    C, H = diffusion.channels, diffusion.image_size
    if hi > lo:
        x = _sample_shard(diffusion, lo, hi, C * H * H, lambda: diffusion.sample(
            conds[lo:hi], None, batch_size=hi - lo, mask=None, min_max_val=min_max_val))
    else:                                                    # more ranks than patches: replicas idle
        x = conds.new_zeros((0, C, H, H), dtype=torch.float32)
        _idle_rank_advances(diffusion, None)
    allx = gather_patches(x.to(torch.float32), P, dtype=gather_dtype)
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
    # The reference decides "all-ones mask -> plain reverse process" on the mask of the WHOLE batch (torch.unique over
    # the batch, ddpm.py:1110-1117).  A shard must not decide it on its own slice: a rank whose images happen to be
    # all-ones would drop to the single-branch path, return another layout and enter the collective with another
    # shape.  The decision is taken once here, on the global masks, and forced into the shard's call; the layout below
    # derives from the same flags on every rank (also an idle one).
    keep = getattr(diffusion, "_all_ones_forced", None)
    diffusion._all_ones_forced = bool(diffusion._all_ones(masks))
    try:
        if hi > lo:
            out = _sample_shard(diffusion, lo, hi, C * H * H, lambda: diffusion.sample(
                cond_img[lo:hi], None if gt is None else gt[lo:hi], batch_size=hi - lo,
                mask=None if masks is None else masks[lo:hi], min_max_val=min_max_val, **sample_kw))
        else:
            _idle_rank_advances(diffusion, masks)
        layout = diffusion.result_layout(masks)
    finally:
        diffusion._all_ones_forced = keep
    as_list, stacked = layout == "list", layout == "stacked"
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


def sample_kmask_sharded(diffusion, cond_img, gt, masks, min_max_val, comm=None, gather_dtype=None):
    """The reference's fusion semantics generalised to K masks, with the K branch-patches of EVERY image spread over the
    ranks (SURVEY 8e "Collective"; ddpm.py:779-810, 955-962, DDIM :1021-1042).  masks: [B, K, H, W].

      1. branch phase: unit u = k * B + b (branch k of image b) -> rank by contiguous blocks; every rank runs its units
         from t = T-1 down to the fusion step with no traffic (all branches of an image share each step's draw, :852-858);
      2. ONE all-gather of [x_t, x0_hat] per unit, taken at t = start_timestep (in ``gather_dtype``, None = fp32);
      3. images -> ranks by contiguous blocks: recomposition (ld_fuse_ddpm_k / ld_fuse_ddim_k) and the remaining
         <= start_timestep joint steps for the rank's images;
      4. ONE all-gather of the finished images.

    Without fusion (``start_intermediate`` off) step 2's gather returns the K branch states and nothing follows
    ([K, B, C, H, W] for DDPM, a K-list for DDIM: what the unsharded loops return).  With K * B < world the surplus ranks
    idle in step 1, with B < world in step 3 (SURVEY 8e "If K*B < G").  The result equals
    ``diffusion.sample(cond_img, gt, mask=masks)`` sample for sample: every unit and image draws its slice of the one noise
    stream.  ``comm``: an ``LdComm`` (the C ABI's ld_allgather) instead of torch.distributed."""
    B, K = int(masks.shape[0]), int(masks.shape[1])
    world, rank = _world_rank(comm)
    branch, fuse, _ = diffusion.kmask_flags(masks)
    assert branch and K >= 2, "sample_kmask_sharded: branch mode with K >= 2 masks [B, K, H, W]"
    U = K * B
    ulo, uhi = shard_bounds(U, world, rank)
    pay, where = diffusion.kmask_branch_units(cond_img, masks, min_max_val, ulo, uhi, gt=gt)
    allpay = gather_patches(pay, U, comm=comm, dtype=gather_dtype)
    if not fuse:
        states = allpay[:, 0].to(torch.float32).reshape(K, B, *allpay.shape[2:])
        out = [states[k] for k in range(K)] if diffusion.is_ddim_sampling else states
    else:
        ilo, ihi = shard_bounds(B, world, rank)
        x = diffusion.kmask_fuse_joint(cond_img, masks, min_max_val, allpay, where, ilo, ihi)
        out = gather_patches(x, B, comm=comm)
    diffusion.advance_call_state(masks)              # what sample() leaves behind, on every rank alike
    return out
