"""World-size-2 test of the patch sharding + single all-gather path on CPU (gloo).
The recomposition formula is checked against the oracle's definition (sum_k x_k * m_k)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from localdiffusion_hallucination_amd import dist as ldist


def test_shard_bounds_cover_and_invert():
    for n in (0, 1, 7, 8, 64, 513):
        for world in (1, 2, 3, 8):
            covered = []
            for r in range(world):
                lo, hi = ldist.shard_bounds(n, world, r)
                assert 0 <= lo <= hi <= n
                covered += list(range(lo, hi))
                for p in range(lo, hi):
                    assert ldist.patch_owner(p, n, world) == r
            assert covered == list(range(n))
            sizes = [ldist.shard_bounds(n, world, r)[1] - ldist.shard_bounds(n, world, r)[0] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_items, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        full = torch.randn(n_items, 3, 4, 4)                  # identical on every rank
        local = ldist.shard_patches(full)
        lo, hi = ldist.shard_bounds(n_items, world, rank)
        assert local.shape[0] == hi - lo
        processed = local * 2.0 + 1.0                          # stands for the per-patch reverse loop
        gathered = ldist.gather_patches(processed, n_items)
        ok = torch.equal(gathered, full * 2.0 + 1.0)
        # recomposition of K=4 band masks (oracle formula; the GPU path uses ld_recompose)
        K = 4
        masks = torch.zeros(K, 1, 4, 4)
        for k in range(K):
            masks[k, :, :, k] = 1.0
        imgs = gathered[: (n_items // K) * K].reshape(-1, K, 3, 4, 4)
        rec = (imgs * (masks[None] >= 1).float()).sum(1)
        ref = torch.stack([torch.stack([(full[i * K + k] * 2 + 1)[:, :, k] for k in range(K)], -1)
                           for i in range(n_items // K)]) if n_items >= K else rec
        ok = ok and torch.allclose(rec, ref)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [8, 7, 1])
def test_gather_world2_gloo(n_items):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    results = dict(q.get(timeout=10) for _ in range(world))
    assert results == {0: True, 1: True}


class _StubDiffusion:
    """Stands for GaussianDiffusion in the control flow of dist.sample_*_sharded: `sample` returns a pure function of
    (the conditioning of each sample, that sample's slice of the portable noise stream), so that the sharded result
    can be compared with the unsharded call -- which is exactly what noise_offset has to guarantee."""
    channels, image_size, is_ddim_sampling, branch_out = 3, 4, False, True

    def __init__(self, fuse=True):
        self.noise_offset, self.fuse = 0, fuse
        self.calls = []

    _all_ones_forced = None

    def _all_ones(self, mask):
        """GaussianDiffusion._all_ones: the reference's 'mask is all ones -> plain reverse process' test (ddpm.py:1110)."""
        if self._all_ones_forced is not None:
            return self._all_ones_forced
        return mask is not None and bool((mask == 1).all())

    def result_layout(self, mask):
        return "plain" if self.fuse else "stacked"

    def sample(self, cond, gt, batch_size=16, mask=None, min_max_val=None, **kw):
        """The all-ones decision changes the RESULT (as it changes the reverse process in the real sampler): a shard
        that took it on its own slice of the masks would differ from the unsharded batch."""
        from localdiffusion_hallucination_amd import rng
        assert cond.shape[0] == batch_size
        self.calls.append((batch_size, self.noise_offset))
        z = torch.from_numpy(rng.randn((batch_size, 3, 4, 4), 10, 0, self.noise_offset))
        x = cond * 2.0 + z + (0.0 if mask is None else mask)
        if self._all_ones(mask):
            x = x * 0.5 - 3.0
        return x if self.fuse else torch.stack([x, -x], 0)


def _sharded_worker(rank, world, port, n_items, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(1)
        cond = torch.randn(n_items, 3, 4, 4)                  # identical on every rank
        masks = (torch.rand(n_items, 1, 4, 4) > 0.5).float()
        masks[n_items // 2:] = 1.0        # the LAST rank's shard is all ones, the batch as a whole is not
        ok = True
        for fuse in (True, False):
            whole = _StubDiffusion(fuse).sample(cond, None, batch_size=n_items, mask=masks)       # the single-GPU answer
            gd = _StubDiffusion(fuse)
            got = ldist.sample_images_sharded(gd, cond, None, masks, (0.0, 2.0))
            lo, hi = ldist.shard_bounds(n_items, world, rank)
            ok = ok and got.shape == whole.shape and torch.equal(got, whole)
            ok = ok and gd.calls == ([(hi - lo, lo * 3 * 4 * 4)] if hi > lo else []) and gd.noise_offset == 0
        # independent patches: the control flow up to the gather (recomposition is the GPU kernel ld_recompose)
        gd = _StubDiffusion(True)
        real = ldist.recompose
        ldist.recompose = lambda patches, m: patches                                              # keep the gathered patches
        try:
            K = 2 if n_items % 2 == 0 else 1
            got = ldist.sample_patches_sharded(gd, cond, (0.0, 2.0), n_items // K, K, None)
        finally:
            ldist.recompose = real
        whole = _StubDiffusion(True).sample(cond, None, batch_size=n_items)
        ok = ok and torch.equal(got.reshape(whole.shape), whole)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [8, 5, 1])
def test_sharded_sampling_equals_the_unsharded_call_world2_gloo(n_items):
    """dist.sample_images_sharded / sample_patches_sharded at world size 2 (ragged and idle-rank cases): every rank
    gets the full batch, equal to the unsharded call sample for sample (noise_offset = first owned sample)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert dict(q.get(timeout=10) for _ in range(world)) == {0: True, 1: True}


def test_recompose_has_no_cpu_fallback():
    with pytest.raises(RuntimeError):
        ldist.recompose(torch.zeros(1, 2, 1, 4, 4), torch.zeros(2, 1, 4, 4))
