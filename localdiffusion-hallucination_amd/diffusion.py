"""Host-side mirror of the sampling half of the reference's ``GaussianDiffusion``
(/root/reference/ddpm.py:496-1125) over the gfx950 HIP kernels.

Kept from the reference: the constructor signature (:497-513), the registered schedule buffers and
their names (:567-615, so ``state_dict`` round-trips), ``sample`` (:1078), ``p_sample_loop``
(:930), ``ddim_sample`` (:980), ``p_sample`` (:841), ``call_classifier`` (:622), and the
attributes ``is_ddim_sampling / image_size / channels / num_timesteps``.

Changed in form (results identical, see tests/): the reference flips entries of ``self.config``
while sampling (:780-781, :1023-1024, :1093-1117); here ``config`` is only read, the phase
BRANCH -> (FUSE) -> JOINT is explicit, and the one flag whose flipped value survives into the next
``sample()`` call (``mask_x``) is carried on the object (``_mask_x_get/_mask_x_set``, golden G14).
No tensor leaves the GPU inside the loop (the reference
ping-pongs ``.cpu()``/``.to(device)`` every step, :700-708, :865-869), the conditioning encoder
runs once per phase instead of once per step (its input is constant over t, :434), both branches
are evaluated as ONE batched denoiser call, and the dead OOD-branch evaluation for non-MRI data
(:704-708) is skipped.  ``np.save`` debugging side effects (:793-794, :866-868) are not reproduced.

Noise: ``noise_source='device'`` draws z_t with the counter-based generator on the GPU
(``ld_randn``); ``'host'`` uploads the same stream from ``rng.py`` (bit-identical to the golden
fixtures); a callable ``f(shape, k) -> tensor`` may be supplied instead.
"""
import time

import torch
from torch import nn

from . import _cabi as cabi
from . import rng, schedule
from .tuning import Tuning, runtime_configured

_REPLACE_OUT = ("mnist", "mvtec", "oct", "imagenet")


class _Pace:
    """Keeps the host at most ``ahead`` (Tuning.sub_ahead) replayed steps ahead of the streams (events, spin-wait on the
    oldest; 0: no limit).  Unthrottled, the host builds and writes packets without a pause until the hardware queue is
    full (~38 steps of 2 x 106 nodes) and the GPU runs ~4 % slower for exactly that long: 20 timed steps 1.519 -> 1.483 ms
    per step with a limit of 2 (1.479 with 1, but with twice the run-to-run spread: one late host wake-up is a GPU
    bubble; 1.53 with 8), 400 steps unchanged (there the full queue paces the host anyway); docs/findings.md 64."""

    def __init__(self, streams, ahead):
        self.streams, self.pending, self.ahead = streams, [], int(ahead)

    def step_enqueued(self):
        if self.ahead <= 0:
            return
        evs = [torch.cuda.Event() for _ in self.streams]
        for e, gs in zip(evs, self.streams):
            e.record(gs)
        self.pending.append(evs)
        if len(self.pending) > self.ahead:
            for e in self.pending.pop(0):
                e.synchronize()


def _align_streams(streams):
    """Every stream waits until all of them have reached this point (events, no host wait)."""
    evs = [torch.cuda.Event() for _ in streams]
    for ev, gs in zip(evs, streams):
        ev.record(gs)
    for i, gs in enumerate(streams):
        for j, ev in enumerate(evs):
            if j != i:
                gs.wait_event(ev)


_SIDE_STREAMS = {}       # (device index, CU-mask kind) -> side streams shared by all samplers of the process (_sub_streams)


def _masked_stream(i, kind):
    """A side stream restricted to a subset of the CUs (``GaussianDiffusion.sub_cu_mask``; DESIGN finding 40):
    ``lo`` (mask bits [128 i, 128 i + 128)), ``xcd`` (bits with (bit mod 8) div 4 == i: the KFD stripes mask bits over
    the XCDs, so this is four whole XCDs per stream) or ``even`` (bit mod 2 == i).  None: an ordinary stream."""
    if not kind:
        return None
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    n, i = 256, i % 2
    pick = {"lo": lambda b: b // 128 == i, "xcd": lambda b: (b % 8) // 4 == i, "even": lambda b: b % 2 == i}[kind]
    words = (C.c_uint32 * (n // 32))()
    for b in range(n):
        if pick(b):
            words[b // 32] |= 1 << (b % 32)
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), C.c_uint32(n // 32), words)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed: {rc}")
    return torch.cuda.ExternalStream(st.value)


class _SubBatches:
    """The joint reverse steps of one batch run as S independent sub-batches, each on its own HIP stream as a
    replayed HIP graph of one step.

    Patches do not interact inside the T-loop, so a batch can be cut anywhere.  A step alternates between
    HBM-bound launches (256^2, 128^2 maps) and latency / L2-bound ones (64^2, 32^2 maps: a few hundred
    workgroups, dependent K-chunk round trips); two sub-batches on two hardware queues fill each other's gaps
    (measured on cfg3: 1.87 -> 1.66 ms per step of 8 patches).  Every sub-batch draws its slice of the batch's
    noise stream (ld_final_step_at), so the samples are those of the unsplit batch up to the tiling-dependent
    summation order of the kernels.  x_t is scattered from / gathered back into the parent plan's ``x_in``."""

    def __init__(self, gd, jp, S, shared_noise=False, masked=(), instance_base=0):
        """``shared_noise``: every sub-batch uses the SAME draw (the OOD and IND branch of the branch phase,
        ddpm.py:852-858) instead of its slice of one draw.  ``masked``: sub-batches whose prediction is replaced
        by the range minimum outside a mask (``set_mask``), i.e. ld_mask_out folded into their final step."""
        self.gd, self.jp, self.S = gd, jp, S
        if not runtime_configured() and not getattr(_SubBatches, "_warned", False):
            _SubBatches._warned = True
            import warnings
            warnings.warn("graph-replay sampling without localdiffusion_hallucination_amd.configure_runtime(): the HIP "
                          "runtime's captured AQL packets make replayed steps 1-5 % slower (call it before the first "
                          "GPU call of the process; INTEGRATION.md)", RuntimeWarning, stacklevel=3)
        B, _, H, W = jp.x_in.shape
        self.b = B // S
        # plans are cached per shape: instances 1..S of this batch size belong to the sub-batch runners
        self.plans = [gd.model.plan(self.b, H, W, table_T=gd.num_timesteps_ori, instance=instance_base + i + 1)
                      for i in range(S)]
        self.streams = gd._sub_streams(S)
        self.shared_noise = shared_noise
        self.masks = {i: torch.ones(self.b, H * W, dtype=torch.float32, device=jp.x_in.device) for i in masked}
        self.graphs = {}
        self.cond_seen = None

    def set_mask(self, mask):
        for m in self.masks.values():
            m.copy_(mask.reshape(m.shape))

    def _step(self, i, st, lo, hi, base):
        gd, sp = self.gd, self.plans[i]
        lib = cabi.lib()
        xa, wf, bf = sp.final
        B_, C_, H_, W_ = sp.model_out.shape
        # the step counter moves at the HEAD of the step (inside ld_step_begin): callers park it one above the step
        # they want to run next
        sp.run_main(st, skip_final=True, step_delta=-1)
        cabi.check(lib.ld_final_step_at(xa.data_ptr(), wf.data_ptr(), bf.data_ptr(), sp.model_out.data_ptr(),
                                        sp.x_in.data_ptr(), None, gd._sched_table().data_ptr(), sp.t_dev.data_ptr(),
                                        lo, hi, cabi.OBJ[gd.objective], gd.noise_seed, base, -1,
                                        gd.noise_offset + (0 if self.shared_noise else i * sp.x_in.numel()),
                                        cabi.ptr(self.masks.get(i)),
                                        B_, H_, W_, wf.shape[1], C_, sp.dt, st), "final_step")

    def _ensure_graph(self, i, gs, lo, hi, base):
        """Capture sub-batch i's step for this noise base if not done yet (call with ``gs`` current).  Returns the
        number of steps run eagerly on the way (1 the first time: lazy hipFuncSetAttribute calls must not
        happen inside a capture)."""
        import ctypes as C
        lib, st = cabi.lib(), gs.cuda_stream
        key = (i, float(lo), float(hi), base, self.gd.noise_seed, self.gd.noise_offset)
        if key in self.graphs:
            return 0
        ran = 0
        if not getattr(self, "_warm", False) and not any(k[0] == i for k in self.graphs):
            self._step(i, st, lo, hi, base)
            ran = 1
        gs.synchronize()
        cabi.check(lib.ld_graph_begin(st), "graph_begin")
        try:
            self._step(i, st, lo, hi, base)          # recorded, not run: the step counter stays put
        finally:
            g = C.c_void_p()
            rc = lib.ld_graph_end(st, C.byref(g))
        cabi.check(rc, "graph_end")
        self.graphs[key] = g
        return ran

    def _ensure_joint_graph(self, lo, hi, base):
        """ONE graph of one step of EVERY sub-batch (Tuning.sub_joint_graph): captured on stream 0, forked to the other
        streams through an event they wait for and joined through events stream 0 waits for -- one host launch per step
        instead of S, the sub-batches start each step together (the in-phase regime of finding 44 by construction) and
        meet again at its end.  Requires the per-sub-batch graphs to exist (their capture ran the lazy set-up eagerly)."""
        import ctypes as C
        lib = cabi.lib()
        key = ("joint", float(lo), float(hi), base, self.gd.noise_seed, self.gd.noise_offset)
        if key in self.graphs:
            return self.graphs[key]
        s0 = self.streams[0]
        for gs in self.streams:
            gs.synchronize()
        fork = torch.cuda.Event()
        joins = [torch.cuda.Event() for _ in self.streams[1:]]
        cabi.check(lib.ld_graph_begin(s0.cuda_stream), "graph_begin")
        try:
            fork.record(s0)
            for gs in self.streams[1:]:
                gs.wait_event(fork)                      # joins the capture
            for i, gs in enumerate(self.streams):
                with torch.cuda.stream(gs):
                    self._step(i, gs.cuda_stream, lo, hi, base)
            for ev, gs in zip(joins, self.streams[1:]):
                ev.record(gs)
                s0.wait_event(ev)
        finally:
            g = C.c_void_p()
            rc = lib.ld_graph_end(s0.cuda_stream, C.byref(g))
        cabi.check(rc, "graph_end")
        self.graphs[key] = g
        return g

    def _trim(self, lo, hi, base):
        """Many different noise bases (a bench that wraps over samples) would pile up graphs: start over."""
        if (0, float(lo), float(hi), base, self.gd.noise_seed, self.gd.noise_offset) in self.graphs or len(self.graphs) < 8 * self.S:
            return
        torch.cuda.synchronize()
        for g in self.graphs.values():
            cabi.lib().ld_graph_destroy(g)
        self.graphs.clear()
        self._warm = True

    def run(self, t_start, n_steps, lo, hi, draw):
        lib, jp, b = cabi.lib(), self.jp, self.b
        cur = torch.cuda.current_stream()
        base = draw + t_start                       # noise stream index of step t is base - t (one draw per t > 0)
        self._trim(lo, hi, base)
        recond = self.cond_seen != getattr(jp, "cond_version", 0)
        self.cond_seen = getattr(jp, "cond_version", 0)
        todo, ex = [n_steps] * self.S, [None] * self.S
        for i, (sp, gs) in enumerate(zip(self.plans, self.streams)):
            gs.wait_stream(cur)
            st = gs.cuda_stream
            with torch.cuda.stream(gs):
                sp.x_in.copy_(jp.x_in[i * b:(i + 1) * b])
                if recond:                          # the parent's conditioning changed: encode this slice of it
                    sp.cond_in.copy_(jp.cond_in[i * b:(i + 1) * b])
                    sp.run_cond_replayed(gs)        # one graph launch per sample (the first sample: eager + capture)
                sp.set_step(t_start + 1)
                todo[i] -= self._ensure_graph(i, gs, lo, hi, base)
                ex[i] = self.graphs[(i, float(lo), float(hi), base, self.gd.noise_seed, self.gd.noise_offset)]
        # interleave the launches so that neither hardware queue runs ahead of the other.  The sub-batches run fastest IN
        # PHASE (the same launch of every sub-batch on the chip at the same time: shared weight blocks in L2, launches
        # that end together; 1.56 ms per step against 1.62-1.67 half a step apart, DESIGN finding 44) and a phase, once
        # taken, persists: line the streams up when they start (their encoders were enqueued one after the other) and
        # again every LD_SUB_RESYNC steps (default 32; 0: never).
        h0 = time.perf_counter()
        tn = self.gd.tuning
        if tn.sub_joint_graph and self.S > 1 and min(todo) == max(todo):
            gj = self._ensure_joint_graph(lo, hi, base)
            s0 = self.streams[0]
            for gs in self.streams[1:]:
                s0.wait_stream(gs)                       # their scatter / encoder precede the forked step
            pace = _Pace(self.streams[:1], tn.sub_ahead)
            lib.ld_range_push(b"steps (one forked graph per step, %d sub-batches)" % self.S)
            for k in range(todo[0]):
                cabi.check(lib.ld_graph_launch(gj, s0.cuda_stream), "graph_launch")
                pace.step_enqueued()
            lib.ld_range_pop()
            for gs in self.streams[1:]:
                gs.wait_stream(s0)
            todo = [0] * self.S
        pace = _Pace(self.streams, tn.sub_ahead)
        lib.ld_range_push(b"steps (graph replay, %d sub-batches)" % self.S)
        for k in range(max(todo)):
            # (and in front of step 1: the host enqueues the streams' first replays one after the other, 0.3 ms apart, so
            # the alignment in front of step 0 finds empty queues and aligns nothing)
            if self.S > 1 and tn.sub_resync > 0 and (k % tn.sub_resync == 0 or k <= tn.sub_resync_early) and k < min(todo):
                _align_streams(self.streams)
            for i, gs in enumerate(self.streams):
                if k < todo[i]:
                    cabi.check(lib.ld_graph_launch(ex[i], gs.cuda_stream), "graph_launch")
            pace.step_enqueued()
        lib.ld_range_pop()
        # host seconds spent enqueueing replays (includes back-pressure once the hardware queue is full)
        self.host_launch_s = getattr(self, "host_launch_s", 0.0) + time.perf_counter() - h0
        self.host_launch_n = getattr(self, "host_launch_n", 0) + sum(todo)
        for i, (sp, gs) in enumerate(zip(self.plans, self.streams)):
            with torch.cuda.stream(gs):
                jp.x_in[i * b:(i + 1) * b].copy_(sp.x_in)
            cur.wait_stream(gs)
        return draw + min(n_steps, t_start)         # steps t > 0 consumed one draw each

    def run_timed(self, t_start, n_steps, lo, hi, draw, acc):
        """bench.py's per-kernel leg in the regime the timed region runs in: sub-batch 0 steps eagerly with
        HIP events around every launch (``acc``, see _Plan.run_main_timed) while the other sub-batches replay
        their graphs beside it."""
        lib, jp, b, gd = cabi.lib(), self.jp, self.b, self.gd
        cur = torch.cuda.current_stream()
        base = draw + t_start
        self._trim(lo, hi, base)
        sched, obj = gd._sched_table(), cabi.OBJ[gd.objective]
        for i, (sp, gs) in enumerate(zip(self.plans, self.streams)):
            gs.wait_stream(cur)
            with torch.cuda.stream(gs):
                sp.x_in.copy_(jp.x_in[i * b:(i + 1) * b])
                sp.set_step(t_start + (1 if i > 0 else 0))       # replayed steps move the counter at their head
        for i in range(1, self.S):
            with torch.cuda.stream(self.streams[i]):
                self._ensure_graph(i, self.streams[i], lo, hi, base)
            g = self.graphs[(i, float(lo), float(hi), base, self.gd.noise_seed, self.gd.noise_offset)]
            for _ in range(n_steps + 2):
                cabi.check(lib.ld_graph_launch(g, self.streams[i].cuda_stream), "graph_launch")
        sp, gs = self.plans[0], self.streams[0]
        z = torch.empty_like(sp.x_in)
        with torch.cuda.stream(gs):
            st = gs.cuda_stream
            for k in range(n_steps):
                sp.set_step(t_start - k)
                sp.run_main_timed(st, acc)
                cabi.check(lib.ld_randn_at(z.data_ptr(), z.numel(), gd.noise_offset, gd.noise_seed, base, -1, sp.t_dev.data_ptr(), st), "randn")
                if 0 in self.masks:
                    cabi.check(lib.ld_mask_out(sp.model_out.data_ptr(), self.masks[0].data_ptr(), lo, sp.x_in.shape[0],
                                               sp.x_in.shape[1], z.shape[2] * z.shape[3], st), "mask_out")
                cabi.check(lib.ld_ddpm_step(sp.x_in.data_ptr(), sp.model_out.data_ptr(), z.data_ptr(), sp.x_in.data_ptr(),
                                            None, sched.data_ptr(), sp.t_dev.data_ptr(), lo, hi, obj, z.numel(), st), "ddpm_step")
        for gs in self.streams:
            cur.wait_stream(gs)
        torch.cuda.synchronize()
        return draw + min(n_steps, t_start)


class _DdimBranches:
    """The DDIM steps before the fusion time with the OOD and the IND branch as two concurrent sub-batches, each a
    replayed HIP graph of one step on its own stream (the DDIM counterpart of ``_SubBatches(shared_noise=True)``).
    A step reads its timestep and its pair of schedule scalars through a device pair counter (``ld_step_begin`` with
    a timestep table, ``ld_ddim_step_at``), so one captured graph serves every pair (ddpm.py:984-986, 1013-1068)."""

    def __init__(self, gd, B, H, W, mask_x, times, pair_rows, shared=True):
        """``shared=True``: the two sub-batches are the OOD and the IND branch of the same B images (same draw per
        pair, mask_x on sub-batch 0).  ``shared=False``: they are the two halves of one single-branch batch of 2B
        samples, each drawing its own slice of the pair's draw."""
        self.gd, self.B, self.mask_x, self.shared = gd, B, mask_x and shared, shared
        dev = gd.device
        self.plans = [gd.model.plan(B, H, W, table_T=gd.num_timesteps_ori, instance=200 + i) for i in range(2)]
        self.streams = gd._sub_streams(2)
        self.idx = [torch.zeros(1, dtype=torch.int32, device=dev) for _ in range(2)]
        self.times = torch.tensor(times, dtype=torch.int32, device=dev)
        self.table = torch.tensor(pair_rows, dtype=torch.float32, device=dev).contiguous()
        self.mask = torch.ones(B, H * W, dtype=torch.float32, device=dev)
        self.z = [torch.zeros(B, gd.channels, H, W, dtype=torch.float32, device=dev) for _ in range(2)]
        self.with_noise = any(r[6] != 0.0 for r in pair_rows)          # eta > 0: sigma * z is not identically zero
        self.graphs = {}

    def _step(self, i, st, lo, hi):
        gd, sp, lib = self.gd, self.plans[i], cabi.lib()
        B, C, H, W = sp.model_out.shape
        n = sp.x_in.numel()
        sp.run_main(st, idx_ptr=self.idx[i].data_ptr(), t_table=self.times.data_ptr())
        if i == 0 and self.mask_x:
            cabi.check(lib.ld_mask_out(sp.model_out.data_ptr(), self.mask.data_ptr(), lo, B, C, H * W, st), "mask_out")
        zp = None
        if self.with_noise:      # branches use the SAME draw (ddpm.py:1020), halves of a batch their slice; draw of pair k is 1 + k
            first = gd.noise_offset + (0 if self.shared else i * n)
            cabi.check(lib.ld_randn_at(self.z[i].data_ptr(), n, first, gd.noise_seed, 1, 1, self.idx[i].data_ptr(), st), "randn")
            zp = self.z[i].data_ptr()
        cabi.check(lib.ld_ddim_step_at(sp.x_in.data_ptr(), sp.model_out.data_ptr(), zp, sp.x_in.data_ptr(), self.table.data_ptr(),
                                       self.idx[i].data_ptr(), lo, hi, cabi.OBJ[gd.objective], n, st), "ddim_step_at")

    @torch.inference_mode()
    def run_timed(self, n_steps, acc, lo, hi, alone=False):
        """bench.py's per-kernel leg for the strided sampler: branch 0's denoiser evaluations run eagerly with HIP events
        around every launch (``acc``, see _Plan.run_main_timed) while branch 1 replays its captured step beside it
        (``alone``: nothing beside it).  Call after ``run`` has captured the graphs; the states are scratch afterwards."""
        lib = cabi.lib()
        cur = torch.cuda.current_stream()
        g1 = next((g for k, g in self.graphs.items() if k[0] == 1), None)
        for gs in self.streams:
            gs.wait_stream(cur)
        if g1 is not None and not alone:
            self.idx[1].fill_(0)
            for _ in range(n_steps + 2):
                cabi.check(lib.ld_graph_launch(g1, self.streams[1].cuda_stream), "graph_launch")
        sp, gs = self.plans[0], self.streams[0]
        with torch.cuda.stream(gs):
            for k in range(n_steps):
                sp.set_step(int(self.times[min(k, len(self.times) - 1)]))
                sp.run_main_timed(gs.cuda_stream, acc)
        for gs in self.streams:
            cur.wait_stream(gs)
        torch.cuda.synchronize()

    def run(self, x_out, x_in, cond_out, cond_in, mask, first_pair, n_steps, lo, hi):
        """Pairs first_pair .. first_pair + n_steps - 1 for both branches; x_out / x_in are updated in place."""
        import ctypes as C
        lib = cabi.lib()
        cur = torch.cuda.current_stream()
        if mask is not None:
            self.mask.copy_(mask.reshape(self.mask.shape))
        todo, ex = [n_steps, n_steps], [None, None]
        for i, (sp, gs, xv, cv) in enumerate(zip(self.plans, self.streams, (x_out, x_in), (cond_out, cond_in))):
            gs.wait_stream(cur)
            st = gs.cuda_stream
            with torch.cuda.stream(gs):
                sp.x_in.copy_(xv)
                sp.cond_in.copy_(cv)
                sp.run_cond_replayed(gs)                     # one graph launch per sample (the first sample: eager + capture)
                self.idx[i].fill_(first_pair - 1)            # the step's first launch advances the pair counter
                key = (i, float(lo), float(hi), gd_key(self.gd), self.shared)
                if key not in self.graphs:
                    self._step(i, st, lo, hi)                # eager first (lazy attribute calls must not be captured)
                    todo[i] -= 1
                    gs.synchronize()
                    cabi.check(lib.ld_graph_begin(st), "graph_begin")
                    try:
                        self._step(i, st, lo, hi)
                    finally:
                        g = C.c_void_p()
                        rc = lib.ld_graph_end(st, C.byref(g))
                    cabi.check(rc, "graph_end")
                    self.graphs[key] = g
                ex[i] = self.graphs[key]
        pace = _Pace(self.streams, self.gd.tuning.sub_ahead)
        for k in range(max(todo)):
            for i, gs in enumerate(self.streams):
                if k < todo[i]:
                    cabi.check(lib.ld_graph_launch(ex[i], gs.cuda_stream), "graph_launch")
            pace.step_enqueued()
        for sp, gs, xv in zip(self.plans, self.streams, (x_out, x_in)):
            with torch.cuda.stream(gs):
                xv.copy_(sp.x_in)
            cur.wait_stream(gs)


def gd_key(gd):
    return (gd.noise_seed, gd.noise_offset, gd.objective)


class GaussianDiffusion(nn.Module):
    def __init__(self, config, model, *, image_size, timesteps=1000, sampling_timesteps=None,
                 objective="pred_v", beta_schedule="sigmoid", schedule_fn_kwargs=dict(),
                 ddim_sampling_eta=0.0, auto_normalize=False, offset_noise_strength=0.0,
                 min_snr_loss_weight=False, min_snr_gamma=5):
        super().__init__()
        assert not (type(self) == GaussianDiffusion and model.channels != model.out_dim)
        assert not model.random_or_learned_sinusoidal_cond
        assert objective in {"pred_noise", "pred_x0", "pred_v"}
        # ddpm.py:619-620: [0,1] images <-> the sampler's [-1,1] range (off for every caller of the reference, test.py:138)
        self.auto_normalize = bool(auto_normalize)
        self.config = config
        self.branch_out = bool(config["branch_out"])
        self.start_intermediate = bool(config["start_intermediate"])
        self.model = model
        self.channels = model.channels
        self.self_condition = model.self_condition
        self.image_size = image_size
        self.objective = objective
        bufs = schedule.make_buffers(timesteps, beta_schedule, objective, min_snr_loss_weight,
                                     min_snr_gamma, **schedule_fn_kwargs)
        for k, v in bufs.items():
            self.register_buffer(k, v)
        self.num_timesteps = int(timesteps)
        self.num_timesteps_ori = int(timesteps)
        self.sampling_timesteps = timesteps if sampling_timesteps is None else sampling_timesteps
        assert self.sampling_timesteps <= timesteps
        self.is_ddim_sampling = self.sampling_timesteps < timesteps
        self.ddim_sampling_eta = ddim_sampling_eta
        self.offset_noise_strength = offset_noise_strength
        self.cnt = -1
        self.instance = 0
        # classifier gate state (ddpm.py:622-625, 883-916).  classifier_flag is never reset by the reference:
        # once one fused prediction is accepted, later sample() calls on the same object skip the scoring.
        self.classifier = None
        self.classifier_flag = 0
        self.pred_cls = 0.0
        self.classifier_calls = 0
        # config['mask_x'] as the reference carries it from one sample() call to the next (it mutates its config dict;
        # here the dict is only read and the carried value lives on the object): cleared by the fusion step
        # (ddpm.py:780-781, 1023-1024) and by the all-ones fallback (:1114), set by a classifier rejection (:907-908),
        # re-armed by sample() only under ood_AD / ood_confidence (:1106-1108).  So with {mask_x: True, ood_AD: False}
        # the reference's SECOND sample() on one object leaves the OOD prediction unmasked -- reproduced (golden G14).
        # ``first_call_semantics = True`` opts out: every call then behaves like the first one on a fresh object.
        self.first_call_semantics = False
        self._mask_x_carried = None               # None: the sampler has not written the flag yet
        self._mask_x_cfg_seen = None
        self._all_ones_forced = None              # dist.py: the all-ones decision of the GLOBAL batch for a shard's call
        # build-specific knobs (additive; defaults reproduce the reference's behaviour)
        # every performance knob in one object (tuning.py): the denoiser's, or defaults + LD_* environment overrides
        self.tuning = getattr(model, "tuning", None) or Tuning.from_env()
        self.noise_source = "device"
        self.fuse_final_step = bool(self.tuning.fused_final_step)
        self.noise_seed = 10                      # torch.manual_seed(10), ddpm.py:934
        # first element of this object's samples inside each draw of the run's noise stream: a rank that owns
        # samples [lo, hi) of a sharded batch sets lo*C*H*W and then draws exactly the values the unsharded batch
        # would have drawn for them (dist.py)
        self.noise_offset = 0
        self.use_graph = False
        # concurrent sub-batches of the joint steps (see _SubBatches); 1 = one batch on the caller's stream
        self.sub_batches = int(self.tuning.sub_batches)
        self.min_sub_batch = int(self.tuning.min_sub_batch)
        self.sub_cu_mask = None                   # experiments: "xcd" / "lo" / "even" CU-masked sub-batch streams (set before sampling)
        self._sched = None
        self._graphs = {}
        self._subs = {}

    # ------------------------------------------------------------------ small API pieces
    def call_classifier(self):
        """ddpm.py:622-625 builds a PatchCore model from anomalib weights the reference does not ship.  Here the
        gate takes any callable ``x0 (B,C,H,W float32 on the device) -> (score, _, _)`` assigned to ``.classifier``
        before sampling; with config['classifier'] on and no callable the sampler refuses to run."""
        if self.config.get("classifier", False) and self.classifier is None:
            raise RuntimeError("config['classifier'] is on: assign a callable x0 -> (score, _, _) to "
                               ".classifier before sampling (the reference's PatchCore weights are not shipped)")

    @property
    def device(self):
        return self.betas.device

    def _st(self):
        return torch.cuda.current_stream().cuda_stream

    def _sched_table(self):
        """[T, 8] fp32 device table of the per-step scalars (layout: LD_SCHED_* in the header)."""
        if self._sched is None or self._sched.device != self.device:
            cols = [self.posterior_mean_coef1, self.posterior_mean_coef2,
                    (0.5 * self.posterior_log_variance_clipped).exp(),          # ddpm.py:853,858
                    self.sqrt_recip_alphas_cumprod, self.sqrt_recipm1_alphas_cumprod,
                    self.sqrt_alphas_cumprod, self.sqrt_one_minus_alphas_cumprod, self.alphas_cumprod]
            self._sched = torch.stack([c.to(torch.float32) for c in cols], dim=1).contiguous()
        return self._sched

    # ------------------------------------------------------------------ noise
    def _noise(self, buf, k, t_dev=None):
        """Fill ``buf`` with draw #k of the run's noise stream."""
        src = self.noise_source
        if callable(src):
            buf.copy_(src(tuple(buf.shape), k).to(buf.device, torch.float32))
        elif src == "host":
            buf.copy_(torch.from_numpy(rng.randn(tuple(buf.shape), self.noise_seed, k, self.noise_offset)))
        elif src == "device":
            cabi.check(cabi.lib().ld_randn_at(buf.data_ptr(), buf.numel(), self.noise_offset, self.noise_seed, k, 0,
                                              None, self._st()), "randn")
        else:
            raise ValueError(f"noise_source {src!r}")

    # ------------------------------------------------------------------ flags
    def _mask_x_get(self):
        """config['mask_x'] as the reference would hold it now (see __init__).  A caller who edits the dict between
        calls wins over the carried value, as in the reference."""
        cfgv = bool(self.config.get("mask_x", False))
        if self.first_call_semantics or self._mask_x_carried is None or cfgv != self._mask_x_cfg_seen:
            self._mask_x_carried = None
            return cfgv
        return self._mask_x_carried

    def _mask_x_set(self, v):
        if not self.first_call_semantics:
            self._mask_x_carried = bool(v)
            self._mask_x_cfg_seen = bool(self.config.get("mask_x", False))

    def reset_call_state(self):
        """Forget what earlier sample() calls left behind (mask_x carry-over, classifier_flag): the object behaves like
        a freshly constructed one again."""
        self._mask_x_carried, self._mask_x_cfg_seen = None, None
        self.classifier_flag = 0

    def advance_call_state(self, mask):
        """Leave behind what a ``sample()`` call with this (GLOBAL) mask leaves behind in the state that outlives a call --
        the call counter and the carried ``mask_x`` (re-armed under ood_AD / ood_confidence, ddpm.py:1106-1108; cleared by
        the all-ones fallback, :1114, and by the fusion step, :780-781 / :1023-1024) -- without sampling.  dist.py calls it
        on a rank whose shard of a sharded call is empty, so that every rank of the NEXT call runs the same reverse
        process (the classifier gate's data-dependent rejection, :907-908, is per rank by nature)."""
        c = self.config
        self.cnt += 1
        if bool(c.get("ood_AD", False)) or bool(c.get("ood_confidence", False)):
            self._mask_x_set(True)
        branch_cfg = bool(c["branch_out"]) or self.branch_out
        fuse_cfg = bool(c["start_intermediate"]) or self.start_intermediate
        if branch_cfg and self._all_ones(mask):
            self._mask_x_set(False)
        elif branch_cfg and fuse_cfg and mask is not None:
            self._mask_x_set(False)

    def _all_ones(self, mask):
        """ddpm.py:1110-1112 (one device sync per call, as there)."""
        if self._all_ones_forced is not None:
            return self._all_ones_forced
        if mask is None or mask.shape[1] != 1:
            return False
        u = torch.unique(mask)
        return len(u) == 1 and float(u[0]) == 1.0

    def _flags(self, mask):
        """Effective (branch, fuse, mask_x) of this call: ddpm.py:1093-1117 without mutating the dict."""
        c = self.config
        branch = bool(c["branch_out"]) or self.branch_out
        fuse = bool(c["start_intermediate"]) or self.start_intermediate
        mask_x = self._mask_x_get() or bool(c.get("ood_AD", False)) or bool(c.get("ood_confidence", False))
        if branch and self._all_ones(mask):               # "Original reverse process as AD is low"
            branch, fuse, mask_x = False, False, False
        return branch, fuse, mask_x

    def result_layout(self, mask):
        """What sample() returns for this mask: "plain" [B,C,H,W]; "stacked" [2,B,C,H,W] (DDPM, branches never fused,
        ddpm.py:964-970 -- also when the all-ones fallback ran a single branch); "list" of two [B,C,H,W] (DDIM that
        never fused, :1069-1075).  dist.py derives the gather layout from it on every rank, idle ones included."""
        branch, fuse, _ = self._flags(mask)
        if self.is_ddim_sampling:
            return "list" if (branch and not fuse) else "plain"
        start_int = bool(self.config["start_intermediate"]) or self.start_intermediate
        return "stacked" if ((not start_int) and self.branch_out) else "plain"

    def _replaced_out(self, mask_x):
        d = self.config["data"]
        return mask_x and any(k in d for k in _REPLACE_OUT) and "mri" not in d

    # ------------------------------------------------------------------ public sampling surface
    @torch.inference_mode()
    def sample(self, cond_img, gt, batch_size=16, return_all_timesteps=False, return_all_outputs=False,
               mask=None, ood_confidence_ad=False, min_max_val=None, instance=0):
        self.instance = instance
        self.cnt += 1
        self.min_max_val = min_max_val
        self.hr = gt
        # the per-call flag handling of ddpm.py:1106-1117 that outlives the call
        c = self.config
        if bool(c.get("ood_AD", False)) or bool(c.get("ood_confidence", False)):
            self._mask_x_set(True)
        branch_cfg = bool(c["branch_out"]) or self.branch_out
        decided_here = branch_cfg and self._all_ones_forced is None
        if decided_here:                                   # one torch.unique (device sync) per call, reused by the loop
            self._all_ones_forced = self._all_ones(mask)
        try:
            if branch_cfg and self._all_ones_forced:
                self._mask_x_set(False)                    # ddpm.py:1114
            shape = (batch_size, self.channels, self.image_size, self.image_size)
            if self.is_ddim_sampling:
                return self.unnormalize(self.ddim_sample(cond_img, mask, min_max_val, shape, return_all_timesteps=return_all_timesteps))
            out = self.p_sample_loop(cond_img, mask, min_max_val, shape, return_all_timesteps=return_all_timesteps,
                                     return_all_outputs=return_all_outputs)
            if return_all_outputs:                         # (ret, x_start_lst, confidence_map), ddpm.py:972-974
                return (self.unnormalize(out[0]),) + tuple(out[1:])
            return self.unnormalize(out)
        finally:
            if decided_here:
                self._all_ones_forced = None

    @torch.inference_mode()
    def p_sample(self, x, mask, min_max_val, cond_img, t: int, x_self_cond=None, draw=None):
        """One ancestral step, single branch (ddpm.py:841-860 non-branch arm): -> (x_{t-1}, x0).
        ``draw``: index of this step's z in the run's noise stream (the reference calls torch.randn_like, i.e. the
        next draw of the global generator); default = a per-object counter, so repeated calls use fresh noise."""
        lib, st = cabi.lib(), self._st()
        B, C, H, W = x.shape
        model_out = self.model(x, cond_img, torch.full((B,), t, device=x.device, dtype=torch.long))
        z = torch.empty_like(x)
        if t > 0:
            if draw is None:
                draw = self._p_sample_draw = getattr(self, "_p_sample_draw", 0) + 1
            self._noise(z, int(draw))
        x_prev, x0 = torch.empty_like(x), torch.empty_like(x)
        row = self._sched_table()[t:t + 1].contiguous()
        # (row mode: the kernel adds the draw iff it gets a noise pointer -- none at t = 0, ddpm.py:857)
        cabi.check(lib.ld_ddpm_step(x.data_ptr(), model_out.data_ptr(), z.data_ptr() if t > 0 else None, x_prev.data_ptr(),
                                    x0.data_ptr(), row.data_ptr(), None, float(min_max_val[0]),
                                    float(min_max_val[1]), cabi.OBJ[self.objective], x.numel(), st), "ddpm_step")
        return x_prev, x0

    # ------------------------------------------------------------------ training-side forward (no gradient)
    def normalize(self, x):
        """ddpm.py:105-106, 619."""
        return x * 2 - 1 if self.auto_normalize else x

    def unnormalize(self, x):
        """ddpm.py:108-109, 620; applied to what the loops return (:972, :1074).  A list result (DDIM branches that were
        never fused) raises as the reference's ``(list + 1)`` does."""
        if not self.auto_normalize:
            return x
        if isinstance(x, (list, tuple)):
            raise TypeError("can only concatenate list (not \"int\") to list")      # what ddpm.py:109 raises on :1069's list
        return (x + 1) * 0.5

    @torch.inference_mode()
    def q_sample(self, x_start, t, noise=None):
        """ddpm.py:1147-1154 with a timestep per sample: ``t`` int64 [B].  ``noise`` None: the next draw of the run's
        noise stream (``noise_source``)."""
        x0 = x_start.to(self.device, torch.float32).contiguous()
        if noise is None:
            noise = torch.empty_like(x0)
            self._train_draw = getattr(self, "_train_draw", -1) + 1
            self._noise(noise, self._train_draw)
        noise = noise.to(self.device, torch.float32).contiguous()
        t32 = t.to(self.device, torch.int32).contiguous()
        out = torch.empty_like(x0)
        cabi.check(cabi.lib().ld_q_sample_t(x0.data_ptr(), noise.data_ptr(), out.data_ptr(), t32.data_ptr(),
                                            self.sqrt_alphas_cumprod.data_ptr(), self.sqrt_one_minus_alphas_cumprod.data_ptr(),
                                            x0.shape[0], x0[0].numel(), self._st()), "q_sample_t")
        return out

    @torch.inference_mode()
    def p_losses(self, x_start, cond_img, t, noise=None, offset_noise_strength=None, per_sample=False):
        """The training loss of one batch WITHOUT a backward pass (ddpm.py:1156-1201): q_sample at the per-sample
        timesteps ``t``, one denoiser evaluation, the objective's target, the per-sample mean squared error times
        ``loss_weight[t]``, the batch mean.  ``noise`` None: draws from the run's noise stream in the reference's order
        (noise, then the [B,C] offset noise when its strength is positive, :1165-1167).  Returns the scalar loss
        (and the per-sample losses with ``per_sample=True``).  Backward / optimiser are out of scope."""
        assert not self.self_condition
        lib, st, dev = cabi.lib(), self._st(), self.device
        x0 = x_start.to(dev, torch.float32).contiguous()
        B = x0.shape[0]
        if noise is None:
            noise = torch.empty_like(x0)
            self._train_draw = getattr(self, "_train_draw", -1) + 1
            self._noise(noise, self._train_draw)
        noise = noise.to(dev, torch.float32).contiguous()
        strength = self.offset_noise_strength if offset_noise_strength is None else offset_noise_strength
        if strength > 0.0:
            off = torch.empty(B, x0.shape[1], dtype=torch.float32, device=dev)
            self._train_draw = getattr(self, "_train_draw", -1) + 1
            self._noise(off, self._train_draw)
            noise = (noise + strength * off[:, :, None, None]).contiguous()
        x = self.q_sample(x0, t, noise)
        model_out = self.model(x, cond_img.to(dev, torch.float32), t.to(dev, torch.long)).contiguous()
        t32 = t.to(dev, torch.int32).contiguous()
        loss = torch.empty(B, dtype=torch.float32, device=dev)
        cabi.check(lib.ld_p_losses(model_out.data_ptr(), x0.data_ptr(), noise.data_ptr(), t32.data_ptr(),
                                   self.sqrt_alphas_cumprod.data_ptr(), self.sqrt_one_minus_alphas_cumprod.data_ptr(),
                                   self.loss_weight.data_ptr(), loss.data_ptr(), B, x0[0].numel(), cabi.OBJ[self.objective], st),
                   "p_losses")
        total = loss.mean()
        return (total, loss) if per_sample else total

    def forward(self, img, cond_img, train, *args, **kwargs):
        """ddpm.py:1203-1214: draws the timesteps with torch's generator as the reference does (``train=False`` re-seeds
        it with 42 first; drawn on the host so that the values do not depend on the device), then ``p_losses``."""
        b, c, h, w = img.shape
        assert h == self.image_size and w == self.image_size, f"height and width of image must be {self.image_size}"
        if not train:
            torch.random.manual_seed(42)
        t = torch.randint(0, self.num_timesteps, (b,)).long()
        return self.p_losses(self.normalize(img), cond_img, t, *args, **kwargs)

    def _sync_model(self):
        """Captured HIP graphs and sub-batch runners hold raw pointers into the denoiser's plans and packed weights:
        when the model dropped those (Unet.invalidate: load_state_dict, .to(), set_compute_dtype) they are destroyed
        here, so a recycled ``id()`` / device pointer can never replay a stale graph and old plans are not pinned."""
        v = getattr(self.model, "_version", 0)
        if v == getattr(self, "_model_version", None):
            return
        self._model_version = v
        if self._graphs or self._subs:
            torch.cuda.synchronize()
        for g in self._graphs.values():
            cabi.lib().ld_graph_destroy(g)
        self._graphs = {}
        for sub in self._subs.values():
            for g in sub.graphs.values():
                cabi.lib().ld_graph_destroy(g)
            sub.graphs = {}
        self._subs = {}

    def _sub_ok(self, b, H, W):
        """Is a sub-batch of ``b`` samples of H x W worth its own stream?  The threshold was measured at 256^2
        (2 patches per sub-batch lose, 4 win); larger maps count by their pixels (one 512^2 image = 4 such patches)."""
        return b * H * W >= self.min_sub_batch * 256 * 256 or b >= 2 * self.min_sub_batch

    def _sub_streams(self, S):
        """The S side streams of the sub-batch runners, shared by every runner AND every sampler object of the process on this
        device: HIP maps streams onto a few hardware queues per process (four by default), so a second ``GaussianDiffusion``
        with side streams of its own finds two of them on ONE queue and its sub-batches serialise (round 6: a cfg5 sampler built
        after a cfg3 sampler in the same process ran 9.6 images/s instead of 15.0).  The pool is keyed by (device, CU-mask kind)."""
        key = (torch.cuda.current_device(), self.sub_cu_mask)
        pool = _SIDE_STREAMS.setdefault(key, [])
        while len(pool) < S:
            pool.append(_masked_stream(len(pool), self.sub_cu_mask) or torch.cuda.Stream())
        return pool[:S]

    def timed_plan(self, jp):
        """The plan whose launches ``run_joint_steps(..., timers=acc)`` times: sub-batch 0 when the joint steps
        of ``jp`` run as concurrent sub-batches, else ``jp`` itself."""
        sub = self._subs.get((id(jp), self.sub_batches))
        return sub.plans[0] if sub is not None else jp

    # ------------------------------------------------------------------ joint reverse steps
    def _will_sub_batch(self, jp, n_steps, x0_buf=None, after=None, timers=None):
        """The condition under which run_joint_steps hands the steps to the concurrent sub-batch runner."""
        S, B_ = self.sub_batches, jp.x_in.shape[0]
        return (S > 1 and B_ % S == 0 and (B_ // S >= self.min_sub_batch or self._sub_ok(B_ // S, *jp.x_in.shape[2:]))
                and self.noise_source == "device" and x0_buf is None and after is None and timers is None and n_steps >= 4
                and not self.use_graph and jp.x_in.shape[1] == jp.model_out.shape[1])

    def encode_cond(self, jp, n_steps, x0_buf=None, after=None):
        """Evaluate the conditioning encoder for the joint steps that follow (``jp.cond_in`` holds the images).  When those
        steps will run as sub-batches, each sub-batch plan encodes its own slice (``_SubBatches.run`` sees the new
        ``cond_version``) and the parent plan's evaluation would be thrown away: it is skipped."""
        if self._will_sub_batch(jp, n_steps, x0_buf, after):
            jp.cond_version = getattr(jp, "cond_version", 0) + 1
        else:
            jp.run_cond(self._st())

    def run_joint_steps(self, jp, t_start, n_steps, lo, hi, z, draw, x0_buf=None, after=None, timers=None):
        """``n_steps`` ancestral steps t_start, t_start-1, ... on plan ``jp`` (x_t lives in
        ``jp.x_in`` and is updated in place; the conditioning features must already be encoded).
        One step = denoiser evaluation (ddpm.py:716) + x0 clamp + posterior mean + sigma*z
        (ddpm.py:817-838, 857-858).  Returns the next draw index.  Used by p_sample_loop and bench.py."""
        self._sync_model()
        lib, st = cabi.lib(), self._st()
        sched = self._sched_table()
        obj = cabi.OBJ[self.objective]
        n = jp.x_in.numel()
        B_ = jp.x_in.shape[0]
        S = self.sub_batches
        if self._will_sub_batch(jp, n_steps, x0_buf, after, timers):
            key = (id(jp), S)
            if key not in self._subs:
                self._subs[key] = _SubBatches(self, jp, S)
            return self._subs[key].run(t_start, n_steps, lo, hi, draw)
        if timers is not None and (id(jp), S) in self._subs and x0_buf is None and after is None:
            return self._subs[(id(jp), S)].run_timed(t_start, n_steps, lo, hi, draw, timers)
        if (self.use_graph and self.noise_source == "device" and x0_buf is None and after is None
                and timers is None and n_steps > 1):
            return self._run_joint_steps_graph(jp, t_start, n_steps, lo, hi, z, draw)
        t = t_start
        # device noise: final_conv, the posterior update and the noise draw of a step are ONE launch (ld_final_step,
        # bitwise the same result as the three calls below)
        fused = (self.noise_source == "device" and timers is None and self.fuse_final_step
                 and jp.x_in.shape[1] == jp.model_out.shape[1])
        if fused:
            xa, wf, bf = jp.final
            B_, C_, H_, W_ = jp.model_out.shape
            for _ in range(n_steps):
                jp.set_step(t)
                jp.run_main(st, skip_final=True)
                k = draw if t > 0 else 0
                cabi.check(lib.ld_final_step_at(xa.data_ptr(), wf.data_ptr(), bf.data_ptr(), jp.model_out.data_ptr(),
                                                jp.x_in.data_ptr(), cabi.ptr(x0_buf), sched.data_ptr(), jp.t_dev.data_ptr(),
                                                lo, hi, obj, self.noise_seed, k, 0, self.noise_offset, None,
                                                B_, H_, W_, wf.shape[1], C_, jp.dt, st), "final_step")
                if t > 0:
                    draw += 1
                if after is not None:
                    after(t)
                t -= 1
            return draw
        for _ in range(n_steps):
            jp.set_step(t)
            if timers is None:
                jp.run_main(st)
            else:
                jp.run_main_timed(st, timers)
            if t > 0:
                self._noise(z, draw)
                draw += 1
            cabi.check(lib.ld_ddpm_step(jp.x_in.data_ptr(), jp.model_out.data_ptr(), z.data_ptr(),
                                        jp.x_in.data_ptr(), cabi.ptr(x0_buf), sched.data_ptr(),
                                        jp.t_dev.data_ptr(), lo, hi, obj, n, st), "ddpm_step")
            if after is not None:
                after(t)
            t -= 1
        return draw

    def _run_joint_steps_graph(self, jp, t_start, n_steps, lo, hi, z, draw):
        """Same steps as the eager loop, replayed from ONE captured HIP graph: the denoiser plan, the
        noise draw, the posterior update and the step-counter decrement all read the timestep
        through ``jp.t_dev``, so the captured launch sequence is identical for every t."""
        import ctypes as C
        lib = cabi.lib()
        sched = self._sched_table()
        obj = cabi.OBJ[self.objective]
        n = jp.x_in.numel()
        base = draw + t_start                       # noise stream index of step t is base - t
        key = (id(jp), float(lo), float(hi), obj, base, z.data_ptr(), self.noise_seed, self.noise_offset)
        cur = torch.cuda.current_stream()
        if getattr(self, "_gstream", None) is None:
            self._gstream = torch.cuda.Stream()
        gs = self._gstream
        gs.wait_stream(cur)
        st = gs.cuda_stream
        first = key not in self._graphs
        t = t_start
        with torch.cuda.stream(gs):
            if first:
                # one eager step first (lazy hipFuncSetAttribute calls etc. must not happen in capture)
                jp.set_step(t)
                jp.run_main(st)
                cabi.check(lib.ld_randn_at(z.data_ptr(), n, self.noise_offset, self.noise_seed, base, -1, jp.t_dev.data_ptr(), st), "randn")
                cabi.check(lib.ld_ddpm_step(jp.x_in.data_ptr(), jp.model_out.data_ptr(), z.data_ptr(), jp.x_in.data_ptr(),
                                            None, sched.data_ptr(), jp.t_dev.data_ptr(), lo, hi, obj, n, st), "ddpm_step")
                t -= 1
                n_steps -= 1
                gs.synchronize()
                cabi.check(lib.ld_graph_begin(st), "graph_begin")
                try:
                    jp.run_main(st)
                    cabi.check(lib.ld_randn_at(z.data_ptr(), n, self.noise_offset, self.noise_seed, base, -1, jp.t_dev.data_ptr(), st), "randn")
                    cabi.check(lib.ld_ddpm_step(jp.x_in.data_ptr(), jp.model_out.data_ptr(), z.data_ptr(),
                                                jp.x_in.data_ptr(), None, sched.data_ptr(), jp.t_dev.data_ptr(),
                                                lo, hi, obj, n, st), "ddpm_step")
                    cabi.check(lib.ld_step_add(jp.t_dev.data_ptr(), -1, st), "step_add")
                finally:
                    ex = C.c_void_p()
                    rc = lib.ld_graph_end(st, C.byref(ex))
                cabi.check(rc, "graph_end")
                self._graphs[key] = ex
            ex = self._graphs[key]
            if n_steps > 0:
                jp.set_step(t)
                for _ in range(n_steps):
                    cabi.check(lib.ld_graph_launch(ex, st), "graph_launch")
        cur.wait_stream(gs)
        return draw + (t_start - t) + n_steps

    # ------------------------------------------------------------------ DDPM loop
    @torch.inference_mode()
    def p_sample_loop(self, cond_img, mask, min_max_val, shape, return_all_timesteps=False,
                      return_all_outputs=False):
        self._sync_model()
        lib, st, dev = cabi.lib(), self._st(), self.device
        B, C, H, W = shape
        HW, n = H * W, B * C * H * W
        lo, hi = float(min_max_val[0]), float(min_max_val[1])
        T = self.num_timesteps
        branch, fuse, mask_x = self._flags(mask)
        obj = cabi.OBJ[self.objective]
        sched = self._sched_table()
        cond = cond_img.to(dev, torch.float32).contiguous()
        if mask is not None:
            mask = mask.to(dev, torch.float32).contiguous()
        start_t = T - 1
        x_T = torch.empty(shape, dtype=torch.float32, device=dev)
        self._noise(x_T, 0)
        if (bool(self.config["start_intermediate"]) or self.start_intermediate) and self.config.get("use_gt", False):
            t0 = int(self.config["use_gt_timestep"])              # ddpm.py:937-944
            hr = self.hr.to(dev, torch.float32).contiguous()
            cabi.check(lib.ld_q_sample(hr.data_ptr(), x_T.data_ptr(), x_T.data_ptr(),
                                       float(self.sqrt_alphas_cumprod[t0]),
                                       float(self.sqrt_one_minus_alphas_cumprod[t0]), n, st), "q_sample")
            start_t = t0 - 1
        z = torch.empty(shape, dtype=torch.float32, device=dev)
        if branch and mask is not None and mask.shape[1] > 1:        # K-mask generalisation
            if return_all_timesteps:
                # as for two branches: ddpm.py:963 stacks `imgs`, which holds per-branch lists for every branch step
                raise TypeError("return_all_timesteps is only defined for the single-branch reverse process "
                                "(the reference's torch.stack(imgs) fails on the per-branch lists, ddpm.py:865,963)")
            return self._p_sample_loop_kmask(cond, mask, lo, hi, shape, x_T, z, start_t, fuse, mask_x,
                                             return_all_outputs=return_all_outputs)
        x0_buf = torch.empty(shape, dtype=torch.float32, device=dev) if return_all_outputs else None
        if return_all_timesteps and branch:
            # ddpm.py:963 stacks `imgs`, which holds [x_out, x_in] lists for every branch step (:865): torch.stack
            # raises on them in the reference as well
            raise TypeError("return_all_timesteps is only defined for the single-branch reverse process "
                            "(the reference's torch.stack(imgs) fails on the per-branch lists, ddpm.py:865,963)")
        hist_x, hist_x0 = [x_T.clone()] if return_all_timesteps else None, []
        x0_pair = [torch.empty(shape, dtype=torch.float32, device=dev) for _ in range(2)] if (return_all_outputs and branch) else None
        draw = 1
        t = start_t
        xs = None
        gate = bool(self.config.get("classifier", False)) and branch and fuse     # ddpm.py:883
        if gate:
            self.call_classifier()
        x_branchout = None
        # ---------------- BRANCH phase
        if branch:
            assert self.objective == "pred_x0", "branch mode exists only for pred_x0 (ddpm.py:739-749)"
            assert mask is not None
            if mask_x:
                assert len(torch.unique((mask >= 1.0).float())) == 2, "mask should be binary"   # ddpm.py:698
            lo_clip = 0.5 if self.config["data"] == "mnist" else 0.95
            cond_out, cond_in = torch.empty_like(cond), torch.empty_like(cond)
            cabi.check(lib.ld_branch_conditions(cond.data_ptr(), mask.data_ptr(), cond_out.data_ptr(),
                                                cond_in.data_ptr(), lo_clip, B, cond.shape[1], HW, st), "branch_conditions")
            replaced = self._replaced_out(mask_x)
            nb = B if replaced else 2 * B
            plan = self.model.plan(nb, H, W, table_T=self.num_timesteps_ori)
            if replaced:
                plan.cond_in.copy_(cond_in)
            else:
                plan.cond_in[:B].copy_(cond_out)
                plan.cond_in[B:].copy_(cond_in)
            plan.run_cond(st)
            # x_out lives in its own buffer when the OOD branch is not evaluated
            x_in_view = plan.x_in if replaced else plan.x_in[B:]
            x_out_view = torch.empty(shape, dtype=torch.float32, device=dev) if replaced else plan.x_in[:B]
            x_out_view.copy_(x_T)
            x_in_view.copy_(x_T)
            mo_in = plan.model_out if replaced else plan.model_out[B:]
            mo_out = cond_out if replaced else plan.model_out[:B]
            # the steps before the fusion time: the OOD and the IND branch as two concurrent sub-batches (same
            # draw for both, ld_mask_out folded into the OOD branch's final step)
            n_plain = (t - int(self.config["start_timestep"])) if fuse else t + 1
            if (not replaced and self.sub_batches > 1 and (B >= self.min_sub_batch or self._sub_ok(B, H, W))
                    and self.noise_source == "device"
                    and not return_all_timesteps and not return_all_outputs and n_plain >= 4
                    and C == plan.model_out.shape[1] and not self.use_graph):
                key = (id(plan), "branch", bool(mask_x))
                if key not in self._subs:
                    self._subs[key] = _SubBatches(self, plan, 2, shared_noise=True, masked=(0,) if mask_x else (),
                                                  instance_base=100)
                if mask_x:
                    self._subs[key].set_mask(mask)
                draw = self._subs[key].run(t, n_plain, lo, hi, draw)
                t -= n_plain
            while t >= 0:
                plan.set_step(t)
                plan.run_main(st)
                if mask_x and not replaced:
                    cabi.check(lib.ld_mask_out(mo_out.data_ptr(), mask.data_ptr(), lo, B, C, HW, st), "mask_out")
                if t > 0:
                    self._noise(z, draw)
                    draw += 1
                if fuse and t <= int(self.config["start_timestep"]):
                    jp = self.model.plan(B, H, W, table_T=self.num_timesteps_ori)
                    x0f = torch.empty(shape, dtype=torch.float32, device=dev)
                    if gate:                               # self.x_branchout, ddpm.py:799
                        m1 = (mask >= 1.0).float()
                        x_branchout = [x_out_view * m1, x_in_view * (1.0 - m1)]
                    cabi.check(lib.ld_fuse_ddpm(x_out_view.data_ptr(), x_in_view.data_ptr(), mo_out.data_ptr(),
                                                mo_in.data_ptr(), mask.data_ptr(), jp.x_in.data_ptr(),
                                                x0f.data_ptr(), lo, hi, B, C, HW, st), "fuse_ddpm")
                    self._mask_x_set(False)                # ddpm.py:781
                    jp.set_step(t)
                    cabi.check(lib.ld_posterior_step(jp.x_in.data_ptr(), x0f.data_ptr(), z.data_ptr(),
                                                     jp.x_in.data_ptr(), sched.data_ptr(), jp.t_dev.data_ptr(), n, st),
                               "posterior_step")
                    if return_all_outputs:
                        hist_x0.append(x0f.cpu())
                    if return_all_timesteps:
                        hist_x.append(jp.x_in.clone())
                    t -= 1
                    branch = False
                    break
                plan.set_step(t)
                for bi, (xv, mv) in enumerate(((x_out_view, mo_out), (x_in_view, mo_in))):
                    cabi.check(lib.ld_ddpm_step(xv.data_ptr(), mv.data_ptr(), z.data_ptr(), xv.data_ptr(),
                                                cabi.ptr(x0_pair[bi]) if x0_pair else None,
                                                sched.data_ptr(), plan.t_dev.data_ptr(), lo, hi, obj, n, st), "ddpm_step")
                if return_all_outputs:                     # branching_out(), ddpm.py:869
                    hist_x0.append([x0_pair[0].cpu(), x0_pair[1].cpu()])
                t -= 1
            if branch:                                     # never fused
                xs = [x_out_view.clone(), x_in_view.clone()]
        # ---------------- JOINT phase
        if xs is None:
            jp = self.model.plan(B, H, W, table_T=self.num_timesteps_ori)
            if t == start_t:                               # no branch phase ran: start from x_T
                jp.x_in.copy_(x_T)
            jp.cond_in.copy_(cond)
            def after(_t):
                if return_all_outputs:
                    hist_x0.append(x0_buf.cpu())
                if return_all_timesteps:
                    hist_x.append(jp.x_in.clone())
            cb = after if (return_all_outputs or return_all_timesteps) else None
            if gate and x_branchout is not None:
                jp.run_cond(st)
                if x0_buf is None:
                    x0_buf = torch.empty(shape, dtype=torch.float32, device=dev)
                self._gated_joint_steps(jp, t, lo, hi, z, draw, x0_buf, x_branchout, cond, cond_out, cond_in,
                                        mask, mask_x, after if (return_all_outputs or return_all_timesteps) else None)
            else:
                self.encode_cond(jp, t + 1, x0_buf, cb)
                self.run_joint_steps(jp, t, t + 1, lo, hi, z, draw, x0_buf=x0_buf, after=cb)
            ret = jp.x_in.clone()
        else:
            ret = xs
        if return_all_timesteps and not isinstance(ret, list):
            ret = torch.stack(hist_x, dim=1)
        start_int = bool(self.config["start_intermediate"]) or self.start_intermediate
        if (not start_int) and self.branch_out:            # ddpm.py:964-970
            ret = torch.stack(ret, dim=0) if isinstance(ret, list) else torch.stack((ret, ret), dim=0)
        if return_all_outputs:
            return ret, hist_x0, []
        return ret

    def _kmask_setup(self, cond, masks, lo, shape, mask_x):
        """Conditioning, plan and state / prediction views of a K-branch run (shared by the DDPM and the DDIM loop)."""
        lib, st, dev = cabi.lib(), self._st(), self.device
        B, C, H, W = shape
        K, HW = masks.shape[1], H * W
        assert self.objective == "pred_x0", "branch mode exists only for pred_x0 (ddpm.py:739-749)"
        m0 = masks[:, 0:1].contiguous()
        if mask_x:
            assert len(torch.unique((m0 >= 1.0).float())) == 2, "mask should be binary"   # ddpm.py:698
        lo_clip = 0.5 if self.config["data"] == "mnist" else 0.95
        cond_k = torch.empty((K,) + tuple(cond.shape), dtype=torch.float32, device=dev)
        cabi.check(lib.ld_branch_conditions_k(cond.data_ptr(), masks.data_ptr(), cond_k.data_ptr(), lo_clip, B,
                                              cond.shape[1], K, HW, st), "branch_conditions_k")
        replaced = self._replaced_out(mask_x)
        nb = (K - 1) * B if replaced else K * B
        plan = self.model.plan(nb, H, W, table_T=self.num_timesteps_ori)
        plan.cond_in.copy_((cond_k[1:] if replaced else cond_k).reshape(nb, *cond.shape[1:]))
        plan.run_cond(st)
        x_first = torch.empty(shape, dtype=torch.float32, device=dev) if replaced else plan.x_in[:B]
        x_rest = plan.x_in if replaced else plan.x_in[B:]
        mo_first = cond_k[0] if replaced else plan.model_out[:B]
        mo_rest = plan.model_out if replaced else plan.model_out[B:]
        return plan, m0, cond_k, replaced, x_first, x_rest, mo_first, mo_rest

    def _p_sample_loop_kmask(self, cond, masks, lo, hi, shape, x_T, z, start_t, fuse, mask_x, return_all_outputs=False):
        """Branch -> fusion -> joint with K >= 2 masks [B,K,H,W] (the reference's loop, ddpm.py:672-708,
        769-810, 852-858, has K = 2).  Branch 0 is the OOD-style branch (hard-masked conditioning; with mask_x its
        prediction is replaced by the range minimum outside m_0, or by its conditioning for the datasets of :704-708),
        branches 1..K-1 are IND-style (conditioning floored at 0.95 / 0.5).  All branches share each step's draw; at
        t <= start_timestep they are recomposed (x0 = clamp(sum_k x0_k m_k), x_t = first non-zero of x_t,k m_k) and the
        remaining steps run on the fused image.  Without fusion the K branch states come back as [K,B,C,H,W].
        ``return_all_outputs``: (ret, x_start_lst, []) -- a K-list of x0 per branch step (ddpm.py:869), the fused /
        single x0 otherwise (:879, 922).  config['classifier']: the joint steps run under the gate of fusion()
        (:883-916) with a K-branch redo.  oracle/diffusion_ref.py::p_sample_loop_kmask is the restatement; for K = 2 and
        m_1 = 1 - (m_0 >= 1) both are bitwise the two-branch path."""
        lib, st, dev = cabi.lib(), self._st(), self.device
        B, C, H, W = shape
        K, HW, n = masks.shape[1], H * W, B * C * H * W
        sched, obj = self._sched_table(), cabi.OBJ[self.objective]
        masks = masks.to(dev, torch.float32).contiguous()
        plan, m0, cond_k, replaced, x_first, x_rest, mo_first, mo_rest = self._kmask_setup(cond, masks, lo, shape, mask_x)
        x_first.copy_(x_T)
        x_rest.copy_(x_T.repeat(K - 1, 1, 1, 1))
        gate = bool(self.config.get("classifier", False)) and fuse
        if gate:
            self.call_classifier()
        hist_x0 = []
        x0_k = [torch.empty(shape, dtype=torch.float32, device=dev) for _ in range(K)] if return_all_outputs else None
        t, draw = start_t, 1
        fused, kept = False, None
        while t >= 0:
            plan.set_step(t)
            plan.run_main(st)
            if mask_x and not replaced:
                cabi.check(lib.ld_mask_out(mo_first.data_ptr(), m0.data_ptr(), lo, B, C, HW, st), "mask_out")
            if t > 0:
                self._noise(z, draw)
                draw += 1
            if fuse and t <= int(self.config["start_timestep"]):
                jp = self.model.plan(B, H, W, table_T=self.num_timesteps_ori)
                x0f = torch.empty(shape, dtype=torch.float32, device=dev)
                if gate:                                       # self.x_branchout (ddpm.py:799), one state per branch
                    bm = (masks >= 1.0).float()
                    kept = [x_first * bm[:, 0:1]] + [x_rest[(k - 1) * B:k * B] * bm[:, k:k + 1] for k in range(1, K)]
                cabi.check(lib.ld_fuse_ddpm_k(x_first.data_ptr(), x_rest.data_ptr(), mo_first.data_ptr(), mo_rest.data_ptr(),
                                              masks.data_ptr(), jp.x_in.data_ptr(), x0f.data_ptr(), lo, hi, B, C, K, HW, st),
                           "fuse_ddpm_k")
                self._mask_x_set(False)                    # ddpm.py:781
                jp.set_step(t)
                cabi.check(lib.ld_posterior_step(jp.x_in.data_ptr(), x0f.data_ptr(), z.data_ptr(), jp.x_in.data_ptr(),
                                                 sched.data_ptr(), jp.t_dev.data_ptr(), n, st), "posterior_step")
                if return_all_outputs:
                    hist_x0.append(x0f.cpu())
                t -= 1
                fused = True
                break
            views = [(x_first, mo_first)] + [(x_rest[(k - 1) * B:k * B], mo_rest[(k - 1) * B:k * B]) for k in range(1, K)]
            for k, (xv, mv) in enumerate(views):               # one shared draw for every branch (ddpm.py:852-858)
                cabi.check(lib.ld_ddpm_step(xv.data_ptr(), mv.data_ptr(), z.data_ptr(), xv.data_ptr(),
                                            cabi.ptr(x0_k[k]) if x0_k else None,
                                            sched.data_ptr(), plan.t_dev.data_ptr(), lo, hi, obj, n, st), "ddpm_step")
            if return_all_outputs:                             # branching_out(), ddpm.py:869
                hist_x0.append([b.cpu() for b in x0_k])
            t -= 1
        if not fused:
            ret = torch.cat([x_first.reshape(1, *shape), x_rest.reshape(K - 1, *shape)], 0).clone()
            return (ret, hist_x0, []) if return_all_outputs else ret
        jp.cond_in.copy_(cond)
        jp.run_cond(st)
        x0_buf = torch.empty(shape, dtype=torch.float32, device=dev) if (return_all_outputs or gate) else None
        after = (lambda _t: hist_x0.append(x0_buf.cpu())) if return_all_outputs else None
        if gate:
            self._gated_joint_steps_k(jp, t, lo, hi, z, draw, x0_buf, kept, cond, cond_k, masks, m0, after)
        else:
            self.run_joint_steps(jp, t, t + 1, lo, hi, z, draw, x0_buf=x0_buf, after=after)
        ret = jp.x_in.clone()
        return (ret, hist_x0, []) if return_all_outputs else ret

    def _gated_joint_steps_k(self, jp, t, lo, hi, z, draw, x0_buf, kept, cond, cond_k, masks, m0, after):
        """``_gated_joint_steps`` for K branches: a rejected joint step (score <= 0, t > 0) is replaced by a K-branch
        evaluation + fusion at the same t from the masked branch states ``kept`` (one per branch, ddpm.py:799), with
        mask_x forced on (:906-908)."""
        lib, st = cabi.lib(), self._st()
        sched, obj = self._sched_table(), cabi.OBJ[self.objective]
        B, C, H, W = jp.x_in.shape
        K, HW, n = masks.shape[1], H * W, jp.x_in.numel()
        bp = None
        while t >= 0:
            jp.set_step(t)
            jp.run_main(st)
            if t > 0:
                self._noise(z, draw)
                draw += 1
            cabi.check(lib.ld_ddpm_step(jp.x_in.data_ptr(), jp.model_out.data_ptr(), z.data_ptr(), jp.x_in.data_ptr(),
                                        x0_buf.data_ptr(), sched.data_ptr(), jp.t_dev.data_ptr(), lo, hi, obj, n, st), "ddpm_step")
            if self.classifier_flag == 0:
                self.pred_cls = float(self.classifier(x0_buf)[0])
                self.classifier_calls += 1
            if self.pred_cls > 0.0 or t == 0:
                self.classifier_flag = 1
            else:
                replaced = self._replaced_out(True)
                nb = (K - 1) * B if replaced else K * B
                if bp is None:
                    bp = self.model.plan(nb, H, W, table_T=self.num_timesteps_ori)
                bp.cond_in.copy_((cond_k[1:] if replaced else cond_k).reshape(nb, *cond.shape[1:]))
                bp.run_cond(st)
                x_first = kept[0] if replaced else bp.x_in[:B]
                x_rest = bp.x_in if replaced else bp.x_in[B:]
                if not replaced:
                    x_first.copy_(kept[0])
                for k in range(1, K):
                    x_rest[(k - 1) * B:k * B].copy_(kept[k])
                mo_first = cond_k[0] if replaced else bp.model_out[:B]
                mo_rest = bp.model_out if replaced else bp.model_out[B:]
                bp.set_step(t)
                bp.run_main(st)
                if not replaced:
                    cabi.check(lib.ld_mask_out(mo_first.data_ptr(), m0.data_ptr(), lo, B, C, HW, st), "mask_out")
                self._noise(z, draw)                        # t > 0 here
                draw += 1
                cabi.check(lib.ld_fuse_ddpm_k(x_first.data_ptr(), x_rest.data_ptr(), mo_first.data_ptr(), mo_rest.data_ptr(),
                                              masks.data_ptr(), jp.x_in.data_ptr(), x0_buf.data_ptr(), lo, hi, B, C, K, HW, st),
                           "fuse_ddpm_k")
                self._mask_x_set(False)
                if bp is jp:                                # plans are cached per batch size: restore the joint one
                    jp.cond_in.copy_(cond)
                    jp.run_cond(st)
                jp.set_step(t)
                cabi.check(lib.ld_posterior_step(jp.x_in.data_ptr(), x0_buf.data_ptr(), z.data_ptr(), jp.x_in.data_ptr(),
                                                 sched.data_ptr(), jp.t_dev.data_ptr(), n, st), "posterior_step")
            if after is not None:
                after(t)
            t -= 1
        return draw

    def _ddim_sample_kmask(self, cond, masks, lo, hi, shape, x_T, fuse, mask_x):
        """ddim_sample with K >= 2 masks [B,K,H,W]: every pair evaluates the K branches as one batch (clip_x_start,
        one shared draw, ddpm.py:1005-1020), the fusion pair (t <= times[-start_timestep-2]) recomposes them with
        ld_fuse_ddim_k, the remaining pairs run on the fused image; never fused -> a list of the K branch states.
        oracle/diffusion_ref.py::ddim_sample_kmask is the restatement (K = 2 is the reference's path bit for bit)."""
        lib, st, dev = cabi.lib(), self._st(), self.device
        B, C, H, W = shape
        K, HW, n = masks.shape[1], H * W, B * C * H * W
        T, S, eta = self.num_timesteps, self.sampling_timesteps, self.ddim_sampling_eta
        obj = cabi.OBJ[self.objective]
        masks = masks.to(dev, torch.float32).contiguous()
        times, pairs = schedule.ddim_time_pairs(T, S)
        t_fuse = times[-int(self.config["start_timestep"]) - 2]                 # ddpm.py:987
        abar, sr_all, srm1_all = self.alphas_cumprod, self.sqrt_recip_alphas_cumprod, self.sqrt_recipm1_alphas_cumprod
        sab_all, s1m_all = self.sqrt_alphas_cumprod, self.sqrt_one_minus_alphas_cumprod

        def scalars(t, t_next):
            a, an = abar[t], abar[t_next]
            sigma = eta * ((1 - a / an) * (1 - an) / (1 - a)).sqrt()             # ddpm.py:1017-1018
            return float(an.sqrt()), float((1 - an - sigma ** 2).sqrt()), float(sigma)

        def step(xv, mv, t, t_next, zz):
            last = 1 if t_next < 0 else 0
            san, c, sigma = (0.0, 0.0, 0.0) if last else scalars(t, t_next)
            cabi.check(lib.ld_ddim_step(xv.data_ptr(), mv.data_ptr(), cabi.ptr(zz), xv.data_ptr(), float(sr_all[t]),
                                        float(srm1_all[t]), float(sab_all[t]), float(s1m_all[t]), san, c, sigma, lo, hi, obj,
                                        last, xv.numel(), st), "ddim_step")
        plan, m0, cond_k, replaced, x_first, x_rest, mo_first, mo_rest = self._kmask_setup(cond, masks, lo, shape, mask_x)
        x_first.copy_(x_T)
        x_rest.copy_(x_T.repeat(K - 1, 1, 1, 1))
        z = torch.zeros(shape, dtype=torch.float32, device=dev)
        draw, idx, jp = 1, 0, None
        while idx < len(pairs):
            t, t_next = pairs[idx]
            plan.set_step(t)
            plan.run_main(st)
            if mask_x and not replaced:
                cabi.check(lib.ld_mask_out(mo_first.data_ptr(), m0.data_ptr(), lo, B, C, HW, st), "mask_out")
            views = [(x_first, mo_first)] + [(x_rest[(k - 1) * B:k * B], mo_rest[(k - 1) * B:k * B]) for k in range(1, K)]
            if t_next < 0:
                for xv, mv in views:
                    step(xv, mv, t, t_next, None)
                idx += 1
                continue
            self._noise(z, draw)
            draw += 1
            if fuse and t <= t_fuse:
                jp = self.model.plan(B, H, W, table_T=self.num_timesteps_ori)
                san, c, sigma = scalars(t, t_next)
                cabi.check(lib.ld_fuse_ddim_k(x_first.data_ptr(), x_rest.data_ptr(), mo_first.data_ptr(), mo_rest.data_ptr(),
                                              masks.data_ptr(), z.data_ptr(), jp.x_in.data_ptr(), float(sr_all[t]),
                                              float(srm1_all[t]), san, c, sigma, lo, hi, B, C, K, HW, st), "fuse_ddim_k")
                self._mask_x_set(False)                    # ddpm.py:1024
                idx += 1
                break
            for xv, mv in views:
                step(xv, mv, t, t_next, z)
            idx += 1
        if jp is None:
            return [x_first.clone()] + [x_rest[(k - 1) * B:k * B].clone() for k in range(1, K)]
        jp.cond_in.copy_(cond)
        jp.run_cond(st)
        while idx < len(pairs):
            t, t_next = pairs[idx]
            jp.set_step(t)
            jp.run_main(st)
            if t_next >= 0:
                self._noise(z, draw)
                draw += 1
            step(jp.x_in, jp.model_out, t, t_next, z if t_next >= 0 else None)
            idx += 1
        return jp.x_in.clone()

    # ------------------------------------------------------------------ the K-mask loop in two halves (dist.sample_kmask_sharded)
    # SURVEY 8e "Collective": in the reference's fusion mode the exchange sits at t = start_timestep (ddpm.py:779-810,
    # :1021-1042): every branch contributes its state x_t and its prediction x0_hat, then the remaining steps run on the
    # recomposed image.  The unit of work before that point is a BRANCH-PATCH u = k * B + b (branch k of image b, the
    # layout of _kmask_setup's plan); after it, an image.  kmask_branch_units runs units [u_lo, u_hi) up to the exchange,
    # kmask_fuse_joint takes the gathered (x_t, x0_hat) of ALL units and finishes images [i_lo, i_hi).  Called with the
    # full ranges the two halves are _p_sample_loop_kmask / _ddim_sample_kmask (tests/test_hip_dist.py).
    def kmask_flags(self, masks):
        """(branch, fuse, mask_x) of a K-mask call as sample() would take them now; the per-call state that outlives the
        call is moved by ``advance_call_state`` (once per rank, shard or no shard)."""
        if bool(self.config.get("classifier", False)):
            raise ValueError("the classifier gate re-branches on a data-dependent score (ddpm.py:883-916): not defined across ranks")
        c = self.config
        mask_x = self._mask_x_get() or bool(c.get("ood_AD", False)) or bool(c.get("ood_confidence", False))
        return bool(c["branch_out"]) or self.branch_out, bool(c["start_intermediate"]) or self.start_intermediate, mask_x

    def _kmask_start(self, shape, gt):
        """x_T of the WHOLE image batch (every branch of image b starts from x_T[b], ddpm.py:955-957) and the first t."""
        lib, st, dev = cabi.lib(), self._st(), self.device
        x_T = torch.empty(shape, dtype=torch.float32, device=dev)
        self._noise(x_T, 0)
        start_t = self.num_timesteps - 1
        if (not self.is_ddim_sampling and (bool(self.config["start_intermediate"]) or self.start_intermediate)
                and self.config.get("use_gt", False)):
            t0 = int(self.config["use_gt_timestep"])              # ddpm.py:937-944
            hr = gt.to(dev, torch.float32).contiguous()
            cabi.check(lib.ld_q_sample(hr.data_ptr(), x_T.data_ptr(), x_T.data_ptr(), float(self.sqrt_alphas_cumprod[t0]),
                                       float(self.sqrt_one_minus_alphas_cumprod[t0]), x_T.numel(), st), "q_sample")
            start_t = t0 - 1
        return x_T, start_t

    def _ddim_scalars(self, t, t_next):
        a, an = self.alphas_cumprod[t], self.alphas_cumprod[t_next]
        sigma = self.ddim_sampling_eta * ((1 - a / an) * (1 - an) / (1 - a)).sqrt()     # ddpm.py:1017-1018
        return float(an.sqrt()), float((1 - an - sigma ** 2).sqrt()), float(sigma)

    @torch.inference_mode()
    def kmask_branch_units(self, cond_img, masks, min_max_val, u_lo, u_hi, gt=None):
        """Branch phase of the K-mask reverse process for units [u_lo, u_hi) (u = k * B + b; masks [B,K,H,W]).  Returns
        ``(payload [n_units, 2, C, H, W], where)``: row 0 the unit's state, row 1 its prediction; with fusion these are
        x_t and x0_hat AT the fusion step (state not yet stepped: ddpm.py:779-797 recomposes x_t and x0_hat of the same t),
        without fusion the final states (row 1 unused).  ``where`` = (t or pair index of the exchange, next draw)."""
        self._sync_model()
        lib, st, dev = cabi.lib(), self._st(), self.device
        branch, fuse, mask_x = self.kmask_flags(masks)
        assert branch and self.objective == "pred_x0", "branch mode exists only for pred_x0 (ddpm.py:739-749)"
        B, K, H, W = masks.shape
        C, HW = self.channels, H * W
        shape = (B, C, H, W)
        lo, hi = float(min_max_val[0]), float(min_max_val[1])
        masks = masks.to(dev, torch.float32).contiguous()
        cond = cond_img.to(dev, torch.float32).contiguous()
        m0 = masks[:, 0:1].contiguous()
        if mask_x:
            assert len(torch.unique((m0 >= 1.0).float())) == 2, "mask should be binary"   # ddpm.py:698
        lo_clip = 0.5 if self.config["data"] == "mnist" else 0.95
        cond_k = torch.empty((K,) + tuple(cond.shape), dtype=torch.float32, device=dev)
        cabi.check(lib.ld_branch_conditions_k(cond.data_ptr(), masks.data_ptr(), cond_k.data_ptr(), lo_clip, B,
                                              cond.shape[1], K, HW, st), "branch_conditions_k")
        replaced = self._replaced_out(mask_x)
        x_T, start_t = self._kmask_start(shape, gt)
        n_units = u_hi - u_lo
        n0 = max(0, min(u_hi, B) - u_lo) if replaced else 0       # leading units whose denoiser evaluation is skipped (:704-708)
        nb = n_units - n0
        pay = torch.empty((n_units, 2, C, H, W), dtype=torch.float32, device=dev)
        # (a rank without units walks the same loop -- it needs `where`, and the loop is where that is decided)
        plan = self.model.plan(nb, H, W, table_T=self.num_timesteps_ori) if nb else None
        ub = [u % B for u in range(u_lo, u_hi)]
        X0 = torch.empty((n0, C, H, W), dtype=torch.float32, device=dev)     # states of the skipped units
        if n0:
            X0.copy_(x_T[u_lo:u_lo + n0])
        if nb:
            idx = torch.tensor(ub[n0:], device=dev)
            plan.x_in.copy_(x_T[idx])
            plan.cond_in.copy_(cond_k.reshape(K * B, *cond.shape[1:])[u_lo + n0:u_hi])
            plan.run_cond(st)
        # runs of units that share k: rows [i0, i1) of the local list <-> images [b0, b1)
        runs, i = [], 0
        while i < n_units:
            k, b0 = (u_lo + i) // B, (u_lo + i) % B
            j = min(n_units, i + B - b0)
            runs.append((k, i, j, b0, b0 + (j - i)))
            i = j

        def rows(i0, i1):
            """(state, prediction) views of local rows [i0, i1) (a run never straddles the skipped / evaluated boundary)."""
            if i1 <= n0:
                return X0[i0:i1], cond_k[0][u_lo + i0:u_lo + i1]
            return plan.x_in[i0 - n0:i1 - n0], plan.model_out[i0 - n0:i1 - n0]

        def evaluate(t):
            if nb:
                plan.set_step(t)
                plan.run_main(st)
            if mask_x and not replaced:
                for k, i0, i1, b0, b1 in runs:
                    if k == 0:
                        cabi.check(lib.ld_mask_out(rows(i0, i1)[1].data_ptr(), m0[b0:b1].data_ptr(), lo, b1 - b0, C, HW, st), "mask_out")

        def finish(where):
            for k, i0, i1, b0, b1 in runs:
                xv, mv = rows(i0, i1)
                pay[i0:i1, 0].copy_(xv)
                pay[i0:i1, 1].copy_(mv)
            return pay, where

        z = torch.zeros(shape, dtype=torch.float32, device=dev)
        sched, obj = self._sched_table(), cabi.OBJ[self.objective]
        if not self.is_ddim_sampling:
            t, draw = start_t, 1
            while t >= 0:
                evaluate(t)
                if t > 0:
                    self._noise(z, draw)
                    draw += 1
                if fuse and t <= int(self.config["start_timestep"]):
                    return finish((t, draw))
                row = sched[t:t + 1].contiguous()
                for k, i0, i1, b0, b1 in runs:               # one shared draw for every branch of an image (ddpm.py:852-858)
                    xv, mv = rows(i0, i1)
                    cabi.check(lib.ld_ddpm_step(xv.data_ptr(), mv.data_ptr(), z[b0:b1].data_ptr() if t > 0 else None, xv.data_ptr(), None,
                                                row.data_ptr(), None, lo, hi, obj, xv.numel(), st), "ddpm_step")
                t -= 1
            return finish((-1, draw))
        T, S = self.num_timesteps, self.sampling_timesteps
        times, pairs = schedule.ddim_time_pairs(T, S)
        t_fuse = times[-int(self.config["start_timestep"]) - 2]                 # ddpm.py:987
        draw, idx = 1, 0
        while idx < len(pairs):
            t, t_next = pairs[idx]
            evaluate(t)
            last = 1 if t_next < 0 else 0
            if not last:
                self._noise(z, draw)
                draw += 1
                if fuse and t <= t_fuse:
                    return finish((idx, draw))
            san, c, sigma = (0.0, 0.0, 0.0) if last else self._ddim_scalars(t, t_next)
            for k, i0, i1, b0, b1 in runs:
                xv, mv = rows(i0, i1)
                cabi.check(lib.ld_ddim_step(xv.data_ptr(), mv.data_ptr(), None if last else z[b0:b1].data_ptr(), xv.data_ptr(),
                                            float(self.sqrt_recip_alphas_cumprod[t]), float(self.sqrt_recipm1_alphas_cumprod[t]),
                                            float(self.sqrt_alphas_cumprod[t]), float(self.sqrt_one_minus_alphas_cumprod[t]),
                                            san, c, sigma, lo, hi, obj, last, xv.numel(), st), "ddim_step")
            idx += 1
        return finish((len(pairs), draw))

    @torch.inference_mode()
    def kmask_fuse_joint(self, cond_img, masks, min_max_val, payload, where, i_lo, i_hi):
        """Second half: ``payload`` [K*B, 2, C, H, W] = every unit's (x_t, x0_hat) at the fusion step ``where`` (all ranks'
        kmask_branch_units results, gathered); recompose images [i_lo, i_hi) (ld_fuse_ddpm_k / ld_fuse_ddim_k:
        ddpm.py:779-810, :1021-1042) and run their remaining steps.  The image's draws are those of the unsharded batch
        (noise stream positioned at image i_lo).  Returns [i_hi - i_lo, C, H, W]."""
        self._sync_model()
        lib, st, dev = cabi.lib(), self._st(), self.device
        B, K, H, W = masks.shape
        C, HW, nl = self.channels, H * W, i_hi - i_lo
        lo, hi = float(min_max_val[0]), float(min_max_val[1])
        if nl <= 0:
            return torch.empty((0, C, H, W), dtype=torch.float32, device=dev)
        masks_l = masks[i_lo:i_hi].to(dev, torch.float32).contiguous()
        cond_l = cond_img[i_lo:i_hi].to(dev, torch.float32).contiguous()
        pk = payload.to(dev, torch.float32).reshape(K, B, 2, C, H, W)[:, i_lo:i_hi]
        x_first, mo_first = pk[0, :, 0].contiguous(), pk[0, :, 1].contiguous()
        x_rest = pk[1:, :, 0].reshape((K - 1) * nl, C, H, W).contiguous()
        mo_rest = pk[1:, :, 1].reshape((K - 1) * nl, C, H, W).contiguous()
        jp = self.model.plan(nl, H, W, table_T=self.num_timesteps_ori)
        shape = (nl, C, H, W)
        z = torch.zeros(shape, dtype=torch.float32, device=dev)
        n = nl * C * HW
        keep = self.noise_offset
        self.noise_offset = keep + i_lo * C * HW
        try:
            if not self.is_ddim_sampling:
                t, draw = where
                if t > 0:
                    self._noise(z, draw - 1)                   # the fusion step's draw (taken before the exchange)
                x0f = torch.empty(shape, dtype=torch.float32, device=dev)
                cabi.check(lib.ld_fuse_ddpm_k(x_first.data_ptr(), x_rest.data_ptr(), mo_first.data_ptr(), mo_rest.data_ptr(),
                                              masks_l.data_ptr(), jp.x_in.data_ptr(), x0f.data_ptr(), lo, hi, nl, C, K, HW, st),
                           "fuse_ddpm_k")
                jp.set_step(t)
                cabi.check(lib.ld_posterior_step(jp.x_in.data_ptr(), x0f.data_ptr(), z.data_ptr(), jp.x_in.data_ptr(),
                                                 self._sched_table().data_ptr(), jp.t_dev.data_ptr(), n, st), "posterior_step")
                t -= 1
                jp.cond_in.copy_(cond_l)
                if t >= 0:
                    self.encode_cond(jp, t + 1)
                    self.run_joint_steps(jp, t, t + 1, lo, hi, z, draw)
                return jp.x_in.clone()
            idx, draw = where
            times, pairs = schedule.ddim_time_pairs(self.num_timesteps, self.sampling_timesteps)
            obj = cabi.OBJ[self.objective]
            t, t_next = pairs[idx]
            self._noise(z, draw - 1)
            san, c, sigma = self._ddim_scalars(t, t_next)
            cabi.check(lib.ld_fuse_ddim_k(x_first.data_ptr(), x_rest.data_ptr(), mo_first.data_ptr(), mo_rest.data_ptr(),
                                          masks_l.data_ptr(), z.data_ptr(), jp.x_in.data_ptr(), float(self.sqrt_recip_alphas_cumprod[t]),
                                          float(self.sqrt_recipm1_alphas_cumprod[t]), san, c, sigma, lo, hi, nl, C, K, HW, st), "fuse_ddim_k")
            idx += 1
            jp.cond_in.copy_(cond_l)
            jp.run_cond(st)
            while idx < len(pairs):
                t, t_next = pairs[idx]
                jp.set_step(t)
                jp.run_main(st)
                last = 1 if t_next < 0 else 0
                if not last:
                    self._noise(z, draw)
                    draw += 1
                san, c, sigma = (0.0, 0.0, 0.0) if last else self._ddim_scalars(t, t_next)
                cabi.check(lib.ld_ddim_step(jp.x_in.data_ptr(), jp.model_out.data_ptr(), None if last else z.data_ptr(), jp.x_in.data_ptr(),
                                            float(self.sqrt_recip_alphas_cumprod[t]), float(self.sqrt_recipm1_alphas_cumprod[t]),
                                            float(self.sqrt_alphas_cumprod[t]), float(self.sqrt_one_minus_alphas_cumprod[t]),
                                            san, c, sigma, lo, hi, obj, last, n, st), "ddim_step")
                idx += 1
            return jp.x_in.clone()
        finally:
            self.noise_offset = keep

    def _gated_joint_steps(self, jp, t, lo, hi, z, draw, x0_buf, x_branchout, cond, cond_out, cond_in, mask,
                           mask_x, after):
        """Joint steps t..0 under the classifier gate (fusion(), ddpm.py:883-916).  Until one fused prediction
        is accepted every joint step's x0 is scored; a rejected one (score <= 0 and t > 0) is discarded and
        replaced by a fresh two-branch evaluation + fusion at the SAME t, started from the masked branch states
        the fusion step kept (ddpm.py:799), with mask_x forced on (:906-908).  Noise draws stay in program
        order: the rejected step's draw, then the redo's."""
        lib, st = cabi.lib(), self._st()
        sched = self._sched_table()
        obj = cabi.OBJ[self.objective]
        B, C, H, W = jp.x_in.shape
        HW, n = H * W, jp.x_in.numel()
        bp = None                                           # branch plan of the redo, built on first rejection
        while t >= 0:
            jp.set_step(t)
            jp.run_main(st)
            if t > 0:
                self._noise(z, draw)
                draw += 1
            cabi.check(lib.ld_ddpm_step(jp.x_in.data_ptr(), jp.model_out.data_ptr(), z.data_ptr(),
                                        jp.x_in.data_ptr(), x0_buf.data_ptr(), sched.data_ptr(),
                                        jp.t_dev.data_ptr(), lo, hi, obj, n, st), "ddpm_step")
            if self.classifier_flag == 0:
                self.pred_cls = float(self.classifier(x0_buf)[0])
                self.classifier_calls += 1
            if self.pred_cls > 0.0 or t == 0:
                self.classifier_flag = 1
            else:
                mask_x = True
                replaced = self._replaced_out(True)
                if bp is None:
                    bp = self.model.plan(B if replaced else 2 * B, H, W, table_T=self.num_timesteps_ori)
                if replaced:
                    bp.cond_in.copy_(cond_in)
                else:
                    bp.cond_in[:B].copy_(cond_out)
                    bp.cond_in[B:].copy_(cond_in)
                bp.run_cond(st)
                x_in_view = bp.x_in if replaced else bp.x_in[B:]
                x_out_view = x_branchout[0] if replaced else bp.x_in[:B]
                if not replaced:
                    x_out_view.copy_(x_branchout[0])
                x_in_view.copy_(x_branchout[1])
                mo_in = bp.model_out if replaced else bp.model_out[B:]
                mo_out = cond_out if replaced else bp.model_out[:B]
                bp.set_step(t)
                bp.run_main(st)
                if not replaced:
                    cabi.check(lib.ld_mask_out(mo_out.data_ptr(), mask.data_ptr(), lo, B, C, HW, st), "mask_out")
                self._noise(z, draw)                        # t > 0 here
                draw += 1
                # the masked states are idempotent under the fusion's own masking, so x_branchout stays as is
                cabi.check(lib.ld_fuse_ddpm(x_out_view.data_ptr(), x_in_view.data_ptr(), mo_out.data_ptr(),
                                            mo_in.data_ptr(), mask.data_ptr(), jp.x_in.data_ptr(),
                                            x0_buf.data_ptr(), lo, hi, B, C, HW, st), "fuse_ddpm")
                self._mask_x_set(False)                     # :908 set it, the redo's fusion step (:781) clears it again
                if bp is jp:                                # plans are cached per batch size: restore the joint one
                    jp.cond_in.copy_(cond)
                    jp.run_cond(st)
                jp.set_step(t)
                cabi.check(lib.ld_posterior_step(jp.x_in.data_ptr(), x0_buf.data_ptr(), z.data_ptr(),
                                                 jp.x_in.data_ptr(), sched.data_ptr(), jp.t_dev.data_ptr(), n, st),
                           "posterior_step")
            if after is not None:
                after(t)
            t -= 1
        return draw

    # ------------------------------------------------------------------ DDIM loop
    @torch.inference_mode()
    def ddim_sample(self, cond_img, mask, min_max_val, shape, return_all_timesteps=False):
        self._sync_model()
        lib, st, dev = cabi.lib(), self._st(), self.device
        B, C, H, W = shape
        HW, n = H * W, B * C * H * W
        lo, hi = float(min_max_val[0]), float(min_max_val[1])
        T, S, eta = self.num_timesteps, self.sampling_timesteps, self.ddim_sampling_eta
        branch, fuse, mask_x = self._flags(mask)
        if branch and mask is not None and mask.shape[1] > 1:        # K-mask generalisation
            if return_all_timesteps:
                raise TypeError("return_all_timesteps is only defined for the single-branch reverse process "
                                "(the reference's torch.stack(imgs) fails on the per-branch lists, ddpm.py:1069-1072)")
            x_T = torch.empty(shape, dtype=torch.float32, device=dev)
            self._noise(x_T, 0)
            return self._ddim_sample_kmask(cond_img.to(dev, torch.float32).contiguous(), mask, lo, hi, shape, x_T, fuse, mask_x)
        obj = cabi.OBJ[self.objective]
        times, pairs = schedule.ddim_time_pairs(T, S)
        t_fuse = times[-int(self.config["start_timestep"]) - 2]                 # ddpm.py:987
        cond = cond_img.to(dev, torch.float32).contiguous()
        if mask is not None:
            mask = mask.to(dev, torch.float32).contiguous()
        x_T = torch.empty(shape, dtype=torch.float32, device=dev)
        self._noise(x_T, 0)
        z = torch.zeros(shape, dtype=torch.float32, device=dev)
        abar = self.alphas_cumprod
        sr_all, srm1_all = self.sqrt_recip_alphas_cumprod, self.sqrt_recipm1_alphas_cumprod
        sab_all, s1m_all = self.sqrt_alphas_cumprod, self.sqrt_one_minus_alphas_cumprod

        def scalars(t, t_next):
            a, an = abar[t], abar[t_next]
            sigma = eta * ((1 - a / an) * (1 - an) / (1 - a)).sqrt()             # ddpm.py:1017-1018
            c = (1 - an - sigma ** 2).sqrt()
            return float(an.sqrt()), float(c), float(sigma)

        def fwd(plan, t):
            plan.set_step(t)
            plan.run_main(st)

        def step(xv, mv, t, t_next, zz):
            last = 1 if t_next < 0 else 0
            san, c, sigma = (0.0, 0.0, 0.0) if last else scalars(t, t_next)
            cabi.check(lib.ld_ddim_step(xv.data_ptr(), mv.data_ptr(), cabi.ptr(zz), xv.data_ptr(),
                                        float(sr_all[t]), float(srm1_all[t]), float(sab_all[t]), float(s1m_all[t]),
                                        san, c, sigma, lo, hi, obj, last, n, st), "ddim_step")

        draw, idx = 1, 0
        xs = None
        if return_all_timesteps and branch:
            # ddpm.py:1072 stacks `imgs`, which holds [x_out, x_in] lists for the branch steps (:1043,1069)
            raise TypeError("return_all_timesteps is only defined for the single-branch reverse process "
                            "(the reference's torch.stack(imgs) fails on the per-branch lists, ddpm.py:1069-1072)")
        hist = [x_T.clone()] if return_all_timesteps else None      # imgs = [img], ddpm.py:993
        if branch:
            assert self.objective == "pred_x0" and mask is not None
            if mask_x:
                assert len(torch.unique((mask >= 1.0).float())) == 2, "mask should be binary"
            lo_clip = 0.5 if self.config["data"] == "mnist" else 0.95
            cond_out, cond_in = torch.empty_like(cond), torch.empty_like(cond)
            cabi.check(lib.ld_branch_conditions(cond.data_ptr(), mask.data_ptr(), cond_out.data_ptr(),
                                                cond_in.data_ptr(), lo_clip, B, cond.shape[1], HW, st), "branch_conditions")
            replaced = self._replaced_out(mask_x)
            nb = B if replaced else 2 * B
            plan = self.model.plan(nb, H, W, table_T=self.num_timesteps_ori)
            if replaced:
                plan.cond_in.copy_(cond_in)
            else:
                plan.cond_in[:B].copy_(cond_out)
                plan.cond_in[B:].copy_(cond_in)
            plan.run_cond(st)
            x_in_view = plan.x_in if replaced else plan.x_in[B:]
            x_out_view = torch.empty(shape, dtype=torch.float32, device=dev) if replaced else plan.x_in[:B]
            x_out_view.copy_(x_T)
            x_in_view.copy_(x_T)
            mo_in = plan.model_out if replaced else plan.model_out[B:]
            mo_out = cond_out if replaced else plan.model_out[:B]
            fused = False
            # the pairs before the fusion time: OOD and IND branch as two concurrent sub-batches (replayed graphs)
            n_plain = 0
            for (t_, tn_) in pairs:
                if tn_ < 0 or (fuse and t_ <= t_fuse):
                    break
                n_plain += 1
            if (not replaced and self.sub_batches > 1 and self.noise_source == "device" and not return_all_timesteps
                    and n_plain >= 4 and not self.use_graph and (B >= self.min_sub_batch or self._sub_ok(B, H, W))):
                key = ("ddim", B, H, W, bool(mask_x), T, S, float(eta))
                if key not in self._subs:
                    rows = []
                    for (t_, tn_) in pairs:
                        san_, c_, sg_ = (0.0, 0.0, 0.0) if tn_ < 0 else scalars(t_, tn_)
                        rows.append([float(sr_all[t_]), float(srm1_all[t_]), float(sab_all[t_]), float(s1m_all[t_]), san_, c_, sg_,
                                     1.0 if tn_ < 0 else 0.0])
                    self._subs[key] = _DdimBranches(self, B, H, W, mask_x, [p_[0] for p_ in pairs], rows)
                self._subs[key].run(x_out_view, x_in_view, cond_out, cond_in, mask, 0, n_plain, lo, hi)
                idx, draw = n_plain, 1 + n_plain
            while idx < len(pairs):
                t, t_next = pairs[idx]
                fwd(plan, t)
                if mask_x and not replaced:
                    cabi.check(lib.ld_mask_out(mo_out.data_ptr(), mask.data_ptr(), lo, B, C, HW, st), "mask_out")
                if t_next < 0:
                    step(x_out_view, mo_out, t, t_next, None)
                    step(x_in_view, mo_in, t, t_next, None)
                    idx += 1
                    continue
                self._noise(z, draw)
                draw += 1
                if fuse and t <= t_fuse:
                    jp = self.model.plan(B, H, W, table_T=self.num_timesteps_ori)
                    san, c, sigma = scalars(t, t_next)
                    cabi.check(lib.ld_fuse_ddim(x_out_view.data_ptr(), x_in_view.data_ptr(), mo_out.data_ptr(),
                                                mo_in.data_ptr(), mask.data_ptr(), z.data_ptr(), jp.x_in.data_ptr(),
                                                float(sr_all[t]), float(srm1_all[t]), san, c, sigma, lo, hi,
                                                B, C, HW, st), "fuse_ddim")
                    self._mask_x_set(False)                # ddpm.py:1024
                    idx += 1
                    fused = True
                    break
                step(x_out_view, mo_out, t, t_next, z)
                step(x_in_view, mo_in, t, t_next, z)
                idx += 1
            if not fused:
                xs = [x_out_view.clone(), x_in_view.clone()]
        if xs is None:
            jp = self.model.plan(B, H, W, table_T=self.num_timesteps_ori)
            if idx == 0:
                jp.x_in.copy_(x_T)
            jp.cond_in.copy_(cond)
            n_left = len(pairs) - idx
            if (self.sub_batches == 2 and B % 2 == 0 and (B // 2 >= self.min_sub_batch or self._sub_ok(B // 2, H, W))
                    and self.noise_source == "device" and not return_all_timesteps and n_left >= 4 and not self.use_graph):
                # single-branch pairs as two concurrent halves of the batch (replayed graphs, sliced draws)
                key = ("ddim-joint", B, H, W, T, S, float(eta))
                if key not in self._subs:
                    rows = []
                    for (t_, tn_) in pairs:
                        san_, c_, sg_ = (0.0, 0.0, 0.0) if tn_ < 0 else scalars(t_, tn_)
                        rows.append([float(sr_all[t_]), float(srm1_all[t_]), float(sab_all[t_]), float(s1m_all[t_]), san_, c_, sg_,
                                     1.0 if tn_ < 0 else 0.0])
                    self._subs[key] = _DdimBranches(self, B // 2, H, W, False, [p_[0] for p_ in pairs], rows, shared=False)
                h = B // 2
                self._subs[key].run(jp.x_in[:h], jp.x_in[h:], cond[:h], cond[h:], None, idx, n_left, lo, hi)
                return jp.x_in.clone()
            jp.run_cond(st)
            while idx < len(pairs):
                t, t_next = pairs[idx]
                fwd(jp, t)
                if t_next >= 0:
                    self._noise(z, draw)
                    draw += 1
                step(jp.x_in, jp.model_out, t, t_next, z if t_next >= 0 else None)
                if hist is not None:
                    hist.append(jp.x_in.clone())
                idx += 1
            return torch.stack(hist, dim=1) if hist is not None else jp.x_in.clone()      # ddpm.py:1072
        return xs
