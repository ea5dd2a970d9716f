#!/usr/bin/env python3
"""Headline benchmark: local patches/sec of the DDPM reverse loop (256x256, T=1000) on N MI355X.

Workload (BASELINE.json configs[2], SURVEY.md 8d "cfg3"): the 4-stage 3-channel denoiser
(12,144,835 parameters, procedural random-init weights), one 256x256 image split into 8 local
patches by 8 vertical band masks => a batch of 8 patch tensors [8,3,256,256] with per-patch
masked conditioning (patch 0: cond*m_0; patches 1..7: cond*clip(m_k, 0.95, 1)), bf16 storage /
fp32 accumulation, ancestral DDPM sampling with T=1000.  Every rank owns one such image
(weak scaling: 8 patches per GPU); patches never talk to each other inside the loop; after the
timed steps one all-gather of the local x tensors + mask recomposition stands for the per-sample
exchange.

A "step" is ONE reverse-diffusion timestep of the whole local patch batch: denoiser evaluation +
clamp + posterior update, exactly the body of GaussianDiffusion.run_joint_steps.  K timed steps
process K/T of a patch, so   value = n_gpus * patches_per_gpu * K / (T * seconds).
The conditioning encoder runs once per sample in the product (its input is constant over t), so
one encoder evaluation is executed INSIDE the timed region.

Prints ONE JSON line on rank 0 (see the task contract), including
  roofline     -- dominant kernel family: algorithmic bytes (or flops) per launch / live per-launch HIP-event time,
                  with the per-family table (launches per step, bytes and flops per launch, average duration) that
                  the fraction can be recomputed from, by hand or from profiles/*_kernel_stats.csv
  cpu_baseline -- the oracle (plain PyTorch fp32 CPU port of the reference) on this box's cores.

N > 1: under torch.distributed.run (WORLD_SIZE set) every process is one rank; called plainly as
`python bench.py --gpus N` the script starts N rank processes itself (before anything touches the GPU) and relays
rank 0's JSON line; it exits non-zero if a rank fails or fewer than N GPUs are visible.
"""
import argparse
import gc
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

T_STEPS = 1000
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_PEAK_TF = {"bf16": 2500.0, "fp16": 2500.0, "fp32": 157.3}   # dense MFMA peaks (same guide)
ALGO_TB_PER_PATCH = 0.7036       # algorithmic bytes per 3x256x256 bf16 patch over T=1000


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--dtype", default=None, choices=["bf16", "fp16", "fp32"],
                    help="storage dtype; default: bf16 for cfg3 (BASELINE configs[2]), fp16 for cfg5 (configs[4])")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--patches", type=int, default=8, help="local patches per GPU (K masks of one image)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-other-dtype", action="store_true", help="skip the short extra run in the other 16-bit storage type")
    ap.add_argument("--workload", default="cfg3", choices=["cfg3", "cfg5"],
                    help="cfg3 (default, the headline metric) | cfg5: BASELINE configs[4] as stated -- one 1x512x512 image per GPU, "
                         "DDIM 50 of 1000 steps, OOD/IND branches with a circular mask, fusion at times[-4]; reports images/s")
    ap.add_argument("--images", type=int, default=None,
                    help="cfg5 only: images in the whole job (default: one per GPU, image-sharded).  Fewer images than ranks: the "
                         "branch-patch units of every image are spread over the ranks (dist.sample_kmask_sharded)")
    ap.add_argument("--no-legs", action="store_true", help="skip the short cfg4-share (64 patches per GPU) and cfg5 legs of the default line")
    ap.add_argument("--weight-split-levels", type=int, default=None,
                    help="two-term (hi + lo) convolution weights on the first N resolution levels (accuracy mode, DESIGN section 2); "
                         "the default line is measured with 0 and reports the cost of 2 in `two_term_weights`")
    ap.add_argument("--graph", type=int, default=-1, help="1: replay the reverse step from a captured HIP graph (default: eager launches)")
    a = ap.parse_args()
    if a.dtype is None:
        a.dtype = "fp16" if a.workload == "cfg5" else "bf16"
    return a


def visible_gpus():
    """GPUs this process may use, counted WITHOUT touching HIP: a process that has initialised the GPU must never be
    re-executed on this pool, and the launcher's children are the ones that should initialise it.  KFD topology nodes
    with SIMDs are GPUs (CPU nodes have simd_count 0); HIP_ / ROCR_ / CUDA_VISIBLE_DEVICES narrow the set the way the
    runtime would."""
    import glob
    n = 0
    for f in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            props = dict(line.split(None, 1) for line in open(f).read().splitlines() if " " in line)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
        except (OSError, ValueError):
            pass
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def _tail(path, lines=25):
    try:
        return "".join(open(path, errors="replace").readlines()[-lines:])
    except OSError:
        return ""


def launch_ranks(a):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (this process makes no GPU call,
    not even a device count through HIP), relay rank 0's stdout.  Every rank's stderr goes to
    <logdir>/bench_rank<r>.err (LD_BENCH_LOG_DIR, default a fresh temp dir); when a rank fails the tail of the FIRST
    failing rank is printed, the others get LD_BENCH_GRACE seconds (default 30) to finish and are then killed by PID;
    LD_BENCH_RANK_TIMEOUT seconds (default 3600) bound the whole run.  Non-zero exit on any failure or timeout."""
    import socket
    import subprocess
    import tempfile
    n_dev = visible_gpus()
    shared = os.environ.get("LD_BENCH_SHARE_GPU") == "1"     # functional test of the N > 1 path on a 1-GPU box (gloo)
    if n_dev < a.gpus and not shared:
        print(f"bench.py: --gpus {a.gpus} but only {n_dev} GPU(s) visible (KFD topology / *_VISIBLE_DEVICES)", file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    logdir = os.environ.get("LD_BENCH_LOG_DIR") or tempfile.mkdtemp(prefix="ld_bench_")
    os.makedirs(logdir, exist_ok=True)
    timeout = float(os.environ.get("LD_BENCH_RANK_TIMEOUT", "3600"))
    grace = float(os.environ.get("LD_BENCH_GRACE", "30"))
    procs, errs = [], []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r % max(1, n_dev) if shared else r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        errs.append(os.path.join(logdir, f"bench_rank{r}.err"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=open(os.path.join(logdir, f"bench_rank{r}.out"), "wb"), stderr=open(errs[-1], "wb")))
    t0 = time.monotonic()
    first_bad, deadline, why = None, t0 + timeout, None
    while any(p.poll() is None for p in procs):
        now = time.monotonic()
        if first_bad is None:
            bad = [r for r, p in enumerate(procs) if p.poll() not in (None, 0)]
            if bad:
                first_bad = bad[0]
                deadline = min(deadline, now + grace)      # the others are most likely stuck in a collective now
        if now > deadline:
            why = (f"rank {first_bad} failed and the others did not finish within {grace:.0f} s" if first_bad is not None
                   else f"timeout: {timeout:.0f} s (LD_BENCH_RANK_TIMEOUT)")
            for p in procs:                                  # exact PIDs of the children this function started
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.05)
    codes = [p.wait() for p in procs]
    sys.stdout.write(open(os.path.join(logdir, "bench_rank0.out"), errors="replace").read())
    sys.stdout.flush()
    if any(codes) or why:
        bad = first_bad if first_bad is not None else next((r for r, c in enumerate(codes) if c), 0)
        print(f"bench.py: rank exit codes {codes}" + (f" ({why})" if why else "") + f"; per-rank stderr in {logdir}", file=sys.stderr)
        print(f"---- tail of {errs[bad]} ----\n{_tail(errs[bad])}", file=sys.stderr)
        return 1
    return 0


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def band_masks(K, H):
    m = torch.zeros(K, 1, H, H)
    wdt = H // K
    for k in range(K):
        m[k, :, :, k * wdt:(k + 1) * wdt] = 1.0
    return m


def patch_conditions(cond, masks):
    """cfg3 per-patch conditioning (ddpm.py:677-688): OOD-style hard mask for patch 0,
    IND-style soft mask (floor 0.95) for the others."""
    out = [cond * masks[0]]
    for k in range(1, masks.shape[0]):
        out.append(cond * torch.clip(masks[k], 0.95, 1.0))
    return torch.cat(out, 0)


def _cpu_steps(cfg, sd, H, seconds_budget, max_steps=40):
    """Oracle reverse steps of ONE patch on the current intra-op pool until the budget is used -> (steps, seconds per step)."""
    from oracle import unet_ref
    from localdiffusion_hallucination_amd import rng
    x = torch.from_numpy(rng.randn((1, cfg.channels, H, H), 3, 0))
    cond = torch.from_numpy(rng.uniform((1, cfg.cond_in_channels, H, H), 3, 1, 0.0, 2.0))
    t = torch.full((1,), 500, dtype=torch.long)
    with torch.no_grad():
        unet_ref.unet_forward(sd, cfg, x, cond, t)        # warm-up
        n, t0 = 0, time.time()
        while True:
            y = unet_ref.unet_forward(sd, cfg, x, cond, t)
            x = 0.9 * x + 0.1 * y.clamp(0, 2)
            n += 1
            if time.time() - t0 > seconds_budget or n >= max_steps:
                break
    return n, (time.time() - t0) / n


def physical_cpus():
    """One logical CPU per physical core among the CPUs this process may use (the lowest id of every SMT sibling set)."""
    try:
        usable = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = list(range(os.cpu_count() or 1))
    prim = []
    for c in usable:
        try:
            sib = open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read().strip()
            first = int(sib.replace("-", ",").split(",")[0])
        except (OSError, ValueError):
            first = c
        if first == c or first not in usable:
            prim.append(c)
    return prim or usable


def cpu_worker_main(argv):
    """``bench.py --cpu-worker H first_core threads seconds``: one worker of the whole-box CPU baseline -- a separate
    process pinned to its own PHYSICAL cores that steps one patch on the oracle and prints {"steps", "s_per_step"}.  Touches
    no GPU."""
    H, first, threads, seconds = int(argv[0]), int(argv[1]), int(argv[2]), float(argv[3])
    try:
        cpus = physical_cpus()
        os.sched_setaffinity(0, set(cpus[first:first + threads]) or set(cpus))
    except (AttributeError, OSError):
        pass
    torch.set_num_threads(threads)
    import localdiffusion_hallucination_amd as ldh
    from localdiffusion_hallucination_amd import weights
    net = ldh.Unet(dim=32, init_dim=32, channels=3, out_dim=3, mode="mvtec")
    sd = {k: torch.from_numpy(v) for k, v in weights.procedural_state_dict(net.cfg, 0).items()}
    n, dt = _cpu_steps(net.cfg, sd, H, seconds)
    print(json.dumps({"steps": n, "s_per_step": dt}))
    return 0


def cpu_baseline(cfg, sd, H, patches, seconds_budget=12.0):
    """The oracle (the CPU restatement of the reference path, oracle/unet_ref.py) on the box's host cores, fp32, a bounded
    sample.  Two figures: ``value`` = the WORKLOAD on the box -- the batch's ``patches`` patches stepped concurrently by
    that many worker processes, each pinned to its own block of CPUs (one patch on 16 threads leaves
    most of a 256-CPU box idle) -- and ``single_patch`` = one patch on the fastest intra-op pool (the latency figure the
    earlier rounds reported)."""
    with torch.no_grad():
        # the GPU box exposes far more logical CPUs than one intra-op pool can use (oversubscribing it makes the oracle
        # ~100x slower): the fastest of a few pool sizes, one timed forward each
        from oracle import unet_ref
        from localdiffusion_hallucination_amd import rng
        x = torch.from_numpy(rng.randn((1, cfg.channels, H, H), 3, 0))
        cond = torch.from_numpy(rng.uniform((1, cfg.cond_in_channels, H, H), 3, 1, 0.0, 2.0))
        t = torch.full((1,), 500, dtype=torch.long)
        try:
            usable = len(os.sched_getaffinity(0))
        except (AttributeError, OSError):
            usable = os.cpu_count()
        best = None
        for nt in sorted({min(usable, k) for k in (8, 16, 32)}):
            torch.set_num_threads(nt)
            unet_ref.unet_forward(sd, cfg, x, cond, t)        # warm-up
            t0 = time.time()
            unet_ref.unet_forward(sd, cfg, x, cond, t)
            dt1 = time.time() - t0
            if best is None or dt1 < best[0]:
                best = (dt1, nt)
            if dt1 > 5.0:
                break
        torch.set_num_threads(best[1])
    n1, dt_single = _cpu_steps(cfg, sd, H, seconds_budget / 2, max_steps=20)
    single = dict(value=1.0 / (T_STEPS * dt_single), unit="patches/s", cores=best[1],
                  sample=f"{n1} consecutive reverse steps of 1 patch on {best[1]} threads (fastest of 8/16/32), {dt_single*1e3:.1f} ms/step")
    # the whole workload: one worker process per patch, each on its own physical cores (spread over the whole box: eight
    # workers packed onto one socket's first 64 CPUs ran 4.6x slower each than one alone)
    cores = len(physical_cpus())
    # at most one worker per 8 physical cores (--patches 64 used to start 64 torch processes); the figure is the
    # BOX's rate -- the sum of the workers' rates, each worker stepping one patch -- whatever the batch size, so it compares
    # across patch counts; `single_patch` below is the one-patch latency figure every round has reported
    workers = max(1, min(patches, cores // 8))
    threads = max(1, min(32, cores // workers))
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(H), str(i * threads), str(threads),
                               str(seconds_budget)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
             for i in range(workers)]
    rates, steps = [], []
    deadline = time.monotonic() + 8 * seconds_budget + 180        # ONE deadline for the whole pool (first import of torch on a fresh box: ~2 min)
    for pr in procs:
        try:
            out, _ = pr.communicate(timeout=max(1.0, deadline - time.monotonic()))
            d = json.loads(out.strip().splitlines()[-1])
            rates.append(1.0 / (T_STEPS * d["s_per_step"]))
            steps.append(d["steps"])
        except Exception:                                       # a worker that died or hung is reported, not hidden
            pr.kill()
    if len(rates) != workers:
        return dict(single, kind="port", host_logical_cpus=os.cpu_count(), host_cpu=cpu_model(),
                    sample=single["sample"] + f"; whole-box leg failed ({len(rates)} of {workers} workers answered)")
    ms = 1e3 / (T_STEPS * (sum(rates) / len(rates)))
    return dict(value=sum(rates), unit="patches/s", cores=threads * workers, kind="port",
                host_logical_cpus=os.cpu_count(), host_usable_cpus=usable, host_cpu=cpu_model(), single_patch=single,
                sample=f"{workers} patches ({cfg.channels}x{H}x{H}, fp32, oracle/unet_ref.py) stepped concurrently by {workers} "
                       f"worker processes x {threads} threads, each pinned to its own physical cores ({cores} on the box): {min(steps)}-{max(steps)} consecutive reverse "
                       f"steps per worker in ~{seconds_budget:.0f} s, mean {ms:.1f} ms/step per patch, extrapolated to T={T_STEPS}; value = the sum of the workers' rates")


def lowp_pin(dtype, dev):
    """What a sample in the bench's storage dtype is worth end to end: cfg2's shape and schedule
    (1x128x128, T = 1000, the fixtures' noise stream) on the CONTRACTIVE procedural denoiser of golden G16 -- generated by
    the real reference, tools/make_goldens.py g16 -- run here in ``dtype`` and compared with the golden's final image.
    On that net the last 100 steps do not amplify a perturbation, so the distance is the implementation's; the same run is
    asserted with bounds in tests/test_hip_lowp_chain.py.  Returns None when the fixture is not in the tree."""
    path = os.path.join(ROOT, "tests", "golden", "g16_cfg2_contractive.npz")
    if not os.path.exists(path):
        return None
    import localdiffusion_hallucination_amd as ldh
    from localdiffusion_hallucination_amd import rng, weights
    g = np.load(path)
    net = ldh.Unet(dim=32, init_dim=32, mode="mri", compute_dtype=dtype)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in weights.procedural_state_dict(net.cfg, 0, final_gain=float(g["final_gain"])).items()})
    net = net.to(dev)
    config = dict(branch_out=False, start_intermediate=False, start_timestep=2, data="mri", mask_x=False,
                  ood_AD=False, ood_confidence=False, classifier=False, use_gt=False)
    gd = ldh.GaussianDiffusion(config, net, image_size=128, timesteps=1000, objective="pred_x0", beta_schedule="sigmoid").to(dev)
    gd.noise_source = "host"
    cond = torch.from_numpy(rng.uniform((1, 1, 128, 128), 5, 1, 0.0, 2.0)).to(dev)
    out = gd.sample(cond, None, batch_size=1, min_max_val=(0.0, 2.0)).cpu().numpy()
    d = np.abs(out - g["final"])
    return {"fixture": "tests/golden/g16_cfg2_contractive.npz (real reference, 1x128x128, T=1000, contractive procedural net)",
            "dtype": dtype, "final_max_abs": float(d.max()), "final_mean_abs": float(d.mean()), "range": [0.0, 2.0],
            "reference_self_distance_max_abs": float(g["self_maxabs_t0"])}


def _family_table(acc, plan, nsteps):
    """Per-launch HIP-event times of ``nsteps`` reverse steps -> per kernel family:
    launches_per_step, bytes_per_launch, flops_per_launch (algorithmic, SURVEY 8d, averaged over the family's launches),
    avg_us (average launch duration) and ms_per_step.  frac of a family = bytes_per_launch / avg_us / peak."""
    fam, per_op = {}, []
    for i, (ms, cnt) in sorted(acc.items()):
        m = plan.meta.get(i, {})
        name = m.get("family", m.get("what", "other").split(" ")[0])
        f = fam.setdefault(name, dict(ms=0.0, launches=0, bytes=0, flops=0))
        f["ms"] += ms
        f["launches"] += cnt
        f["bytes"] += m.get("bytes", 0) * cnt
        f["flops"] += m.get("flops", 0) * cnt
        per_op.append((i, m, 1e3 * ms / max(cnt, 1)))
    table = {}
    for name, f in sorted(fam.items(), key=lambda kv: -kv[1]["ms"]):
        n = max(1, f["launches"])
        table[name] = {"launches_per_step": f["launches"] / nsteps, "bytes_per_launch": round(f["bytes"] / n),
                       "flops_per_launch": round(f["flops"] / n), "avg_us": round(1e3 * f["ms"] / n, 3),
                       "ms_per_step": round(f["ms"] / nsteps, 4)}
    return table, per_op


def _price(row, peak_tf):
    """A family row -> the roofline that bounds it (the larger of its HBM and MFMA fractions) + GB/s, TFLOP/s."""
    gbs = row["bytes_per_launch"] / max(row["avg_us"], 1e-9) / 1e3
    tfs = row["flops_per_launch"] / max(row["avg_us"], 1e-9) / 1e6
    hbm_frac, mfma_frac = gbs / HBM_PEAK_GBS, tfs / peak_tf
    if mfma_frac > hbm_frac:        # a kernel is priced against the roofline that bounds it
        return {"bound": "mfma", "achieved": tfs, "peak": peak_tf, "unit": "TFLOP/s", "frac": mfma_frac}, gbs, tfs
    return {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_frac}, gbs, tfs


def _pmc_traffic(name, solo, tag=None):
    """HBM bytes per launch of family ``name`` from the newest committed rocprofv3 --pmc passes of the SAME regime
    (profiles/*_pmc_traffic.json: the default two-sub-batch regime; *_s1_pmc_traffic.json: one batch on one stream).
    Not live: counters need their own profiler passes (tools/profile_round.sh)."""
    # tag: the workload the passes were collected on (None: cfg3, the headline; "cfg5" / "p64": profiles/*_<tag>_pmc_traffic.json)
    def ok(f):
        if not f.endswith("_pmc_traffic.json"):
            return False
        stem = f[:-len("_pmc_traffic.json")]
        is_solo = stem.endswith("_s1")
        stem = stem[:-3] if is_solo else stem
        ftag = next((t for t in ("cfg5", "p64") if stem.endswith("_" + t)), None)
        return is_solo == solo and ftag == tag
    files = [f for f in os.listdir(os.path.join(ROOT, "profiles")) if ok(f)]
    for cand in sorted(files, reverse=True):
        try:
            t = json.load(open(os.path.join(ROOT, "profiles", cand))).get(name, {}).get("hbm_bytes_per_launch")
        except Exception:
            t = None
        if t is not None:
            return t, "profiles/" + cand
    return None, None


def _leg_block(table, per_op, plan, regime, peak_tf, H, solo, conv_sel=None, traffic_tag=None):
    """One regime's roofline block: the dominant family priced per launch + the table it is recomputed from."""
    name = next(iter(table))                                 # the family with the most time per step
    roof, gbs, tfs = _price(table[name], peak_tf)
    step_ms = sum(r["ms_per_step"] for r in table.values())
    traffic, src = _pmc_traffic(name, solo, traffic_tag)
    roof.update({"kernel": name, "regime": regime, "launch_batch": int(plan.x_in.shape[0]), "traffic": traffic,
                 "traffic_source": src, "avg_launch_us": table[name]["avg_us"], "bytes_per_launch": table[name]["bytes_per_launch"],
                 "flops_per_launch": table[name]["flops_per_launch"], "launches_per_step": table[name]["launches_per_step"],
                 "algorithmic_GBps": gbs, "algorithmic_TFLOPps": tfs, "share_of_step": table[name]["ms_per_step"] / max(step_ms, 1e-9),
                 "step_ms_sum_of_kernels": round(step_ms, 4)})
    if conv_sel is not None:
        # the north star's "ResBlock conv path": the C=32 3x3 convolutions at full resolution (SURVEY 8a census,
        # first two lines), priced against the HBM roofline with their algorithmic bytes
        sel = [(m, us) for _, m, us in per_op if conv_sel(m)]
        us_total, b_total = sum(us for _, us in sel), sum(m.get("bytes", 0) for m, _ in sel)
        roof["resblock_conv_path"] = {"launches_per_step": len(sel), "ms_per_step": round(us_total / 1e3, 4), "bytes_per_step": b_total,
                                      "algorithmic_GBps": b_total / max(us_total, 1e-9) / 1e3,
                                      "hbm_frac": b_total / max(us_total, 1e-9) / 1e3 / HBM_PEAK_GBS}
    roof["families"] = table
    return roof


def conv_path_chip_leg(gd, sub, conv_sel, reps=40):
    """What the CHIP moves for the ResBlock conv path in the timed regime.  ``hbm_frac`` prices ONE sub-batch's
    launches while the other sub-batch's launches share the chip; here the conv-path launches of BOTH sub-batches
    (12 each at cfg3, behind the statistics reset they depend on) are captured as one graph per stream and replayed
    ``reps`` times concurrently, in phase, as the step replays them: (bytes of both sequences) / wall time."""
    import ctypes as C
    lib = cabi_mod().lib()
    graphs, nbytes = [], 0
    for sp, gs in zip(sub.plans, sub.streams):
        idx = [i for i in sorted(sp.meta) if conv_sel(sp.meta[i])]
        nbytes += sum(sp.meta[i].get("bytes", 0) for i in idx)
        st = gs.cuda_stream
        with torch.cuda.stream(gs):
            gs.synchronize()
            cabi_mod().check(lib.ld_graph_begin(st), "graph_begin")
            try:
                cabi_mod().check(lib.ld_step_begin_film(*sp._begin_args, sp._t_dev_ptr, 0, None, None, *sp._film_args, st), "step_begin")
                for i in idx:
                    sp.ops_main[i](st)
            finally:
                g = C.c_void_p()
                rc = lib.ld_graph_end(st, C.byref(g))
            cabi_mod().check(rc, "graph_end")
            graphs.append(g)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    cur = torch.cuda.current_stream()
    best = None
    for _ in range(3):
        for gs in sub.streams:
            gs.wait_stream(cur)
        ev0.record(sub.streams[0])
        for gs in sub.streams[1:]:
            gs.wait_event(ev0)
        for _r in range(reps):
            for g, gs in zip(graphs, sub.streams):
                cabi_mod().check(lib.ld_graph_launch(g, gs.cuda_stream), "graph_launch")
        for gs in sub.streams[1:]:
            sub.streams[0].wait_stream(gs)
        ev1.record(sub.streams[0])
        torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / reps
        best = ms if best is None else min(best, ms)
    for g in graphs:
        lib.ld_graph_destroy(g)
    return {"chip_hbm_frac": nbytes / (best * 1e-3) / 1e9 / HBM_PEAK_GBS, "chip_ms_per_step": round(best, 4), "chip_bytes_per_step": nbytes,
            "chip_method": f"the conv-path launches of all {len(graphs)} sub-batches (+ their statistics reset) as one graph per stream, "
                           f"replayed {reps}x concurrently and in phase (best of 3): bytes of all sequences / wall"}


def cabi_mod():
    from localdiffusion_hallucination_amd import _cabi
    return _cabi


def roofline_leg(gd, jp, tp_value, dtype, H, lo, hi, z, nsteps=5, traffic_tag=None):
    """Per-launch timing (hipExtLaunchKernelGGL start/stop events on the launch stream = rocprofv3's kernel
    durations) of a few reverse steps.  The block describes the regime the TIMED region runs in:
      timed -- the batch as concurrent sub-batches on two streams (replayed step graphs); sub-batch 0 is stepped
               eagerly with events around every launch while the other sub-batch's graphs share the chip.  `frac`,
               `kernel`, `avg_launch_us`, `families` are this regime's; `traffic` comes from the committed --pmc passes
               of the same regime; `rocprofv3 --kernel-trace --stats -- python bench.py` reproduces the durations
               (profiles/*_kernel_stats.csv; the profiler's interception makes the streams overlap somewhat less);
      solo  -- (under `roofline.solo`) the whole local batch as ONE batch on one stream, every launch alone on the chip:
               what `LD_SUB_BATCHES=1 rocprofv3 --kernel-trace --stats` reproduces (profiles/*_s1_kernel_stats.csv).
    When the batch is not split (one sub-batch), the timed regime IS the solo regime and there is no `solo` block."""
    torch.cuda.synchronize()
    jp.run_cond(torch.cuda.current_stream().cuda_stream)      # the solo leg evaluates the parent plan itself
    peak_tf = MFMA_PEAK_TF[dtype]
    legs = {}
    split = gd.timed_plan(jp) is not jp
    for leg in (("solo", "in_situ") if split else ("solo",)):
        acc = {}
        keep = gd.sub_batches
        if leg == "solo":
            gd.sub_batches = 1
        try:
            gd.run_joint_steps(jp, 500, nsteps, lo, hi, z, 1, timers=acc)
            plan = gd.timed_plan(jp) if leg == "in_situ" else jp
        finally:
            gd.sub_batches = keep
        torch.cuda.synchronize()
        table, per_op = _family_table(acc, plan, nsteps)
        legs[leg] = (table, per_op, plan)
    if os.environ.get("LD_BENCH_OPS"):
        with open(os.environ["LD_BENCH_OPS"], "w") as f:
            for leg, (_, ops, _) in legs.items():
                f.write(f"# {leg}\n")
                for i, m, us in ops:
                    f.write(f"{i:4d} {m.get('family', '?'):22s} {m.get('what', '?'):38s} {m.get('shape', ''):20s} "
                            f"{us:9.1f} us  {m.get('bytes', 0) / max(us, 1e-9) / 1e3:8.1f} GB/s  "
                            f"{m.get('flops', 0) / max(us, 1e-9) / 1e6:8.1f} TF/s\n")

    def conv_sel(m):
        return m.get("family", "").startswith("conv3x3") and m.get("shape", "").endswith(f"@{H}x{H}")

    B_ = int(jp.x_in.shape[0])
    t1, ops1, plan1 = legs["solo"]
    solo = _leg_block(t1, ops1, plan1, f"solo: one batch of {B_} on one stream, every launch alone on the chip", peak_tf, H, True, conv_sel, traffic_tag)
    if "in_situ" in legs:
        t2, ops2, plan2 = legs["in_situ"]
        b2 = int(plan2.x_in.shape[0])
        roof = _leg_block(t2, ops2, plan2, f"timed: {gd.sub_batches} concurrent sub-batches of {b2} (replayed step graphs); "
                          f"sub-batch 0's launches timed while the other sub-batch shares the chip", peak_tf, H, False, conv_sel, traffic_tag)
        roof["concurrent_streams"] = gd.sub_batches
        sub = gd._subs.get((id(jp), gd.sub_batches))
        if sub is not None:
            roof["resblock_conv_path"].update(conv_path_chip_leg(gd, sub, conv_sel))
        roof["solo"] = solo
    else:
        roof = solo
        roof["regime"] = f"timed = solo: one batch of {B_} on one stream"
    # whole path: 0.7036 TB of algorithmic traffic per 256^2 patch over T=1000 -- still divided by the
    # figure that INCLUDES the conditioning encoder (34 MB per forward), which the product evaluates once per sample
    roof["path_frac"] = tp_value * ALGO_TB_PER_PATCH * 1e3 / HBM_PEAK_GBS
    return roof


def cfg5_run(dtype, steps, rank, world, dev, dist, images=None, want_roofline=True):
    """BASELINE.json configs[4] as stated: 1-channel 512x512, T=1000 strided to S=50 DDIM steps (eta 0), OOD / IND
    branches with a circular OOD mask (radius 64 at the centre), fusion at times[-4], fp16 unless --dtype says
    otherwise.  A "step" is one DDIM step of an image (both branches before the fusion time, one after); `value` =
    images/s of whole samples.  Returns the JSON object of the line (every rank computes it, rank 0 prints it).

    ``images`` None (default): one image per GPU -- images shard across ranks with no traffic until the final gather
    (weak scaling).  ``images`` < world: the job's images are FEWER than the ranks, so whole images cannot fill the node;
    the K = 2 branch-patches of every image are spread instead (dist.sample_kmask_sharded: units over ranks -> ONE
    all-gather of [x_t, x0_hat] at the fusion step -> images over ranks -> one final gather; SURVEY 8e "If K*B < G")."""
    import localdiffusion_hallucination_amd as ldh
    from localdiffusion_hallucination_amd import rng, weights
    from localdiffusion_hallucination_amd import dist as ldist
    H, T, S = 512, T_STEPS, 50
    net = ldh.Unet(dim=32, init_dim=32, mode="mri", compute_dtype=dtype)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in weights.procedural_state_dict(net.cfg, 0).items()})
    net = net.to(dev)
    config = dict(branch_out=True, start_intermediate=True, start_timestep=2, data="mri", mask_x=True, mask_cond=False,
                  ood_AD=True, ood_confidence=False, classifier=False, use_gt=False, use_gt_timestep=100)
    gd = ldh.GaussianDiffusion(config, net, image_size=H, timesteps=T, objective="pred_x0", beta_schedule="sigmoid",
                               sampling_timesteps=S).to(dev)
    gd.noise_source = "device"
    unit_sharded = images is not None and images < world
    n_img = images if unit_sharded else 1
    gd.noise_offset = 0 if unit_sharded else rank * H * H
    yy, xx = np.mgrid[0:H, 0:H]
    mask = torch.from_numpy((((yy - H / 2) ** 2 + (xx - H / 2) ** 2) <= 64 ** 2).astype(np.float32))[None, None].to(dev)
    if unit_sharded:                # the job's images, identical on every rank; masks in the K-mask form [B, 2, H, W] = [m, 1 - (m >= 1)]
        cond = torch.cat([torch.from_numpy(rng.uniform((1, 1, H, H), 200 + i, 1, 0.0, 2.0)) for i in range(n_img)], 0).to(dev)
        mask = torch.cat([mask, 1.0 - (mask >= 1).float()], 1).expand(n_img, 2, H, H).contiguous()
    else:
        cond = torch.from_numpy(rng.uniform((1, 1, H, H), 200 + rank, 1, 0.0, 2.0)).to(dev)

    def one_sample():
        if unit_sharded:
            return ldist.sample_kmask_sharded(gd, cond, None, mask, (0.0, 2.0))
        return gd.sample(cond, None, batch_size=1, mask=mask, min_max_val=(0.0, 2.0))

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()
    samples = max(1, (steps + S - 1) // S)
    out_img = one_sample()          # warm-up sample (plans, packing)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(samples):
        out_img = one_sample()
    if world > 1 and not unit_sharded:
        gathered = torch.empty(world, 1, H, H, device=dev)
        dist.all_gather_into_tensor(gathered, out_img.contiguous())
    sync_all()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    assert torch.isfinite(out_img).all() and float(out_img.min()) >= 0.0 and float(out_img.max()) <= 2.0
    if rank == 0 and os.environ.get("LD_BENCH_DUMP"):         # tests: the finished image(s) of the last timed sample
        np.save(os.environ["LD_BENCH_DUMP"], (gathered if (world > 1 and not unit_sharded) else out_img).cpu().numpy())
    roof = None
    if rank == 0 and want_roofline and not unit_sharded:
        # per-launch table of the branch phase (47 of the 50 pairs): branch 0 timed launch by launch while branch 1's
        # replayed graph shares the chip (the timed regime), and alone on the chip (`solo`)
        from localdiffusion_hallucination_amd.diffusion import _DdimBranches
        db = next((v for v in gd._subs.values() if isinstance(v, _DdimBranches)), None)
        if db is not None:
            peak_tf, blocks = MFMA_PEAK_TF[dtype], {}
            for leg, alone in (("timed", False), ("solo", True)):
                acc = {}
                db.run_timed(5, acc, 0.0, 2.0, alone=alone)
                table, per_op = _family_table(acc, db.plans[0], 5)
                regime = ("timed: the OOD and the IND branch of one 1x512x512 image as two concurrent sub-batches (replayed step graphs); "
                          "branch 0's launches timed while branch 1 shares the chip") if not alone else \
                         "solo: one branch of one image on one stream, every launch alone on the chip"
                blocks[leg] = _leg_block(table, per_op, db.plans[0], regime, peak_tf, H, alone, traffic_tag="cfg5")
                if os.environ.get("LD_BENCH_OPS"):
                    with open(os.environ["LD_BENCH_OPS"], "a" if leg == "solo" else "w") as f:
                        f.write(f"# {leg}\n")
                        for i, m, us in per_op:
                            f.write(f"{i:4d} {m.get('family', '?'):22s} {m.get('what', '?'):38s} {m.get('shape', ''):20s} {us:9.1f} us\n")
            roof = blocks["timed"]
            roof["solo"] = blocks["solo"]
    n_job = n_img if unit_sharded else world
    return {
        "metric": "images/sec (cfg5: 512^2, DDIM 50 of 1000, branch + fusion)", "value": n_job * samples / elapsed, "unit": "images/s",
        "n_gpus": world, "steps": samples * S, "warmup": S, "ms_per_step": 1e3 * elapsed / (samples * S),
        "higher_is_better": True, "scaling": "strong" if unit_sharded else "weak", "vs_baseline": None, "dtype": dtype,
        "data": "synthetic (portable-RNG conditioning image and x_T, procedural random-init weights)",
        "config": {"workload": "cfg5: one 1x512x512 image per GPU, 4-stage dim-32 conditional UNet, T=1000 / DDIM S=50 (eta 0), "
                               "OOD + IND branches (circular mask r=64), fusion at times[-4], full attention over 4096 tokens",
                   "images_per_gpu": 1 if not unit_sharded else None, "images_in_job": n_job, "sampling_timesteps": S,
                   "parallelism": (f"branch-patch units of {n_img} image(s) over {world} ranks: one all-gather of [x_t, x0_hat] at the fusion step, "
                                   f"images over ranks, one final gather (dist.sample_kmask_sharded)" if unit_sharded
                                   else f"image-sharded x{world}, one all-gather per sample")
                                  + (" [LD_BENCH_SHARE_GPU: ranks share one GPU over gloo -- functional test, not a measurement]"
                                     if os.environ.get("LD_BENCH_SHARE_GPU") == "1" else ""),
                   "denoiser_evaluations_per_image": 2 * (S - 3) + 3},
        **({"roofline": roof} if roof is not None else {})}


def bench_cfg5(a, rank, world, dev, dist):
    out = cfg5_run(a.dtype, a.steps, rank, world, dev, dist, images=a.images, want_roofline=not a.no_roofline)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _compact_leg(line):
    """A whole bench line -> the short form the default line carries for the other BASELINE configs: throughput, step time
    and the dominant kernel family with its roofline fraction (no per-family table)."""
    keep = {k: line[k] for k in ("metric", "value", "unit", "steps", "ms_per_step", "dtype") if k in line}
    r = line.get("roofline")
    if r:
        keep["dominant"] = {k: r.get(k) for k in ("kernel", "bound", "frac", "achieved", "unit", "avg_launch_us", "launches_per_step",
                                                  "share_of_step", "traffic", "traffic_source", "bytes_per_launch", "flops_per_launch")}
        if "path_frac" in r:
            keep["path_frac"] = r["path_frac"]
        if "resblock_conv_path" in r:
            keep["resblock_conv_path"] = {k: v for k, v in r["resblock_conv_path"].items() if k in ("hbm_frac", "chip_hbm_frac", "ms_per_step")}
    return keep


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-worker":
        sys.exit(cpu_worker_main(sys.argv[2:]))
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(launch_ranks(a))
    rank = int(os.environ.get("RANK", "0"))
    # fault injection for the launcher's tests (tests/test_bench_cli.py): before anything touches the GPU
    if os.environ.get("LD_BENCH_FAIL_RANK") in (str(rank), "all"):
        print(f"bench.py: injected failure on rank {rank} (LD_BENCH_FAIL_RANK)", file=sys.stderr)
        sys.exit(3)
    if os.environ.get("LD_BENCH_HANG_RANK") in (str(rank), "all"):
        time.sleep(1e6)
    import localdiffusion_hallucination_amd as ldh
    ldh.configure_runtime()                              # before the first GPU call: graph packet capture off (finding 47)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("LD_BENCH_SHARE_GPU") == "1":
        local %= max(1, torch.cuda.device_count())
    if world != a.gpus:
        print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    if world > 1:
        import torch.distributed as dist
        # RCCL ("nccl" IS RCCL on ROCm).  LD_BENCH_SHARE_GPU=1 (test only: several ranks on one GPU, which RCCL refuses)
        # switches to gloo so that the N > 1 control flow can be exercised on a 1-GPU box; the JSON says so.
        if os.environ.get("LD_BENCH_SHARE_GPU") == "1":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from localdiffusion_hallucination_amd import _cabi as cabi, rng, weights

    if a.workload == "cfg5":
        return bench_cfg5(a, rank, world, dev, dist if world > 1 else None)
    H, P = a.size, a.patches
    net = ldh.Unet(dim=32, init_dim=32, channels=3, out_dim=3, mode="mvtec", compute_dtype=a.dtype)
    sd = {k: torch.from_numpy(v) for k, v in weights.procedural_state_dict(net.cfg, 0).items()}
    net.load_state_dict(sd)
    net = net.to(dev)
    if a.weight_split_levels is None:                       # default: the Unet's own default (0, or LD_WEIGHT_SPLIT_LEVELS)
        a.weight_split_levels = net.weight_split_levels
    net.set_weight_split_levels(a.weight_split_levels)
    config = dict(branch_out=False, start_intermediate=False, start_timestep=2, data="mvtec", mask_x=False,
                  ood_AD=False, ood_confidence=False, classifier=False, use_gt=False)
    gd = ldh.GaussianDiffusion(config, net, image_size=H, timesteps=T_STEPS, objective="pred_x0",
                               beta_schedule="sigmoid").to(dev)
    gd.noise_source = "device"
    gd.noise_offset = rank * a.patches * 3 * a.size * a.size     # every rank draws its own slice of the job's noise stream
    gd.sub_cu_mask = os.environ.get("LD_BENCH_CU_MASK") or None      # experiments: "xcd" = each sub-batch stream owns four XCDs
    gd.use_graph = a.graph == 1          # HIP-graph replay of the reverse step (measured: no gain, the step is GPU-bound)

    cond_img = torch.from_numpy(rng.uniform((1, 3, H, H), 100 + rank, 1, 0.0, 2.0))
    lib, st = cabi.lib(), torch.cuda.current_stream().cuda_stream
    lo, hi = 0.0, 2.0
    inputs = {}

    def workload(Pn):
        """band masks + per-patch conditioning of ``Pn`` local patches (built once per patch count)"""
        if Pn not in inputs:
            m = band_masks(Pn, H)
            inputs[Pn] = (m, patch_conditions(cond_img, m).to(dev))
        return inputs[Pn]

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def measure(steps, warmup, Pn=P):
        """W untimed + K timed reverse steps of ``Pn`` local patches in the model's current storage dtype -> (seconds, plan, z)."""
        masks, conds = workload(Pn)
        jp = net.plan(Pn, H, H, table_T=T_STEPS)
        jp.cond_in.copy_(conds)
        x_T = torch.empty(Pn, 3, H, H, device=dev)
        gd._noise(x_T, 0)
        jp.x_in.copy_(x_T)
        z = torch.empty_like(x_T)
        # inputs of the per-sample exchange are resident before the timed region starts (the masks are an input of the path;
        # uploading them inside it was 2 MB of pageable host-to-device copy per measurement: ~0.4 ms, 1.5 % of a 20-step run)
        img = torch.empty(1, 3, H, H, device=dev)
        mk = masks.reshape(Pn, H * H).to(dev)
        gathered = torch.empty(world * Pn, 3, H, H, device=dev) if world > 1 else None
        # (the host paces a 27 ms timed region of the default run two steps ahead of the GPU: a collector pause in it is a GPU bubble.
        #  Collected HERE, in front of the warm-up: tens of milliseconds of an idle GPU between the warm-up and the timed region
        #  would start the region on a chip that has dropped its clocks)
        gc.collect()
        gc.disable()
        try:                                                            # (an exception in the region must not leave the collector off)
            # warm-up (untimed): encoder + W steps, then straight into the timed region
            gd.encode_cond(jp, warmup)
            draw = gd.run_joint_steps(jp, T_STEPS - 1, warmup, lo, hi, z, 1)
            sync_all()
            _tl = [] if os.environ.get("LD_BENCH_TIMELINE") else None
            t0 = time.perf_counter()
            t_start = T_STEPS - 1 - warmup
            done, new_sample, parent_encoded = 0, True, False
            while done < steps:                                         # wrap to a new sample after T steps
                chunk = min(steps - done, t_start + 1)
                # the conditioning encoder runs once per SAMPLE, inside the timed region.  Whether a chunk's steps run as
                # sub-batches is decided per chunk (run_joint_steps: >= 4 steps): a short chunk at a wrap runs on the
                # parent plan, whose features must then have been encoded as well.
                if new_sample:
                    gd.encode_cond(jp, chunk)
                    parent_encoded = not gd._will_sub_batch(jp, chunk)
                elif not gd._will_sub_batch(jp, chunk) and not parent_encoded:
                    jp.run_cond(st)
                    parent_encoded = True
                if _tl is not None:
                    _tl.append(("encode enqueued", time.perf_counter() - t0))
                draw = gd.run_joint_steps(jp, t_start, chunk, lo, hi, z, draw)
                if _tl is not None:
                    _tl.append((f"{chunk} steps enqueued", time.perf_counter() - t0))
                done += chunk
                t_start -= chunk
                new_sample = t_start < 0
                if new_sample:
                    t_start = T_STEPS - 1
            # per-sample exchange: all-gather the local patches and recompose by the masks
            xl = jp.x_in
            if world > 1:
                dist.all_gather_into_tensor(gathered, xl.contiguous())
                xl = gathered[rank * Pn:(rank + 1) * Pn]
            cabi.check(lib.ld_recompose(xl.contiguous().data_ptr(), mk.data_ptr(), img.data_ptr(), 1, Pn, 3, H * H, st), "recompose")
            if _tl is not None:
                _tl.append(("recompose enqueued", time.perf_counter() - t0))
            sync_all()
            elapsed = time.perf_counter() - t0
        finally:
            gc.enable()
        if _tl is not None:                                         # LD_BENCH_TIMELINE=1: host clock inside the timed region (stderr)
            _tl.append(("synchronised", elapsed))
            print("timed region, host clock [ms]: " + ", ".join(f"{k} {1e3 * v:.3f}" for k, v in _tl), file=sys.stderr)
        if world > 1:
            tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
        assert torch.isfinite(img).all(), "non-finite output"
        return elapsed, jp, z

    elapsed, jp, z = measure(a.steps, a.warmup)
    for _rep in range(int(os.environ.get("LD_BENCH_REPEAT", "1")) - 1):     # diagnosis only: the line reports the FIRST measurement
        e_again, _, _ = measure(a.steps, a.warmup)
        print(f"measurement {_rep + 2}: {1e3 * e_again / a.steps:.4f} ms per step (first: {1e3 * elapsed / a.steps:.4f})", file=sys.stderr)

    value = world * P * a.steps / (T_STEPS * elapsed)
    out = {
        "metric": "local patches/sec (256^2, T=1000)", "value": value, "unit": "patches/s", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype,
        "data": "synthetic (portable-RNG conditioning image and x_T, procedural random-init weights)",
        # what the timed region holds (rounds 1-3: also the 2 MB pageable upload of the recomposition masks, ~0.4 ms; since
        # round 4 every input of the path is resident in HBM before the timer starts, as the bench contract says)
        "timed_region": "v2: inputs resident; one encoder evaluation + K reverse steps + all-gather/recompose",
        "config": {"workload": f"cfg3: {P} local patches (vertical band masks) of one 3x{H}x{H} image per GPU, "
                               f"4-stage dim-32 conditional UNet (12.1M params), DDPM T={T_STEPS}, pred_x0, sigmoid schedule",
                   "patches_per_gpu": P, "image": [3, H, H], "timesteps": T_STEPS,
                   "parallelism": f"patch-sharded x{world}, one all-gather per sample"
                                  + (" [LD_BENCH_SHARE_GPU: ranks share one GPU over gloo -- functional test, not a measurement]"
                                     if os.environ.get("LD_BENCH_SHARE_GPU") == "1" else ""),
                   # how the timed steps were issued: as replayed HIP graphs of one step per sub-batch (the default
                   # regime, _SubBatches), as ONE replayed graph of the whole batch (--graph 1), or eagerly
                   "step_graph_replay": bool(gd.timed_plan(jp) is not jp or gd.use_graph),
                   "whole_batch_graph_flag": bool(gd.use_graph),
                   "concurrent_sub_batches": (gd.sub_batches if gd.timed_plan(jp) is not jp else 1),
                   "weight_split_levels": a.weight_split_levels},
    }

    if rank == 0 and not a.no_roofline:
        out["roofline"] = roofline_leg(gd, jp, tp_value=value / world, dtype=a.dtype, H=H, lo=lo, hi=hi, z=z)
    if world == 1 and a.dtype in ("bf16", "fp16") and not a.no_other_dtype:
        # the same workload in the other 16-bit storage type (BASELINE configs[2] says bf16, the north star fp16): a
        # shorter timed run after the contract's measurement, reported beside it
        other = "fp16" if a.dtype == "bf16" else "bf16"
        net.set_compute_dtype(other)
        k2 = min(a.steps, 200)
        e2, _, _ = measure(k2, min(a.warmup, 10))
        out["other_dtype"] = {"dtype": other, "value": P * k2 / (T_STEPS * e2), "unit": "patches/s", "steps": k2,
                              "ms_per_step": 1e3 * e2 / k2}
        net.set_compute_dtype(a.dtype)
        if a.weight_split_levels == 0:
            # the accuracy mode: two-term weights on the full- and half-resolution layers (the weight rounding of those
            # layers is what separates a 16-bit chain from the fp32 reference: 4x closer at t = 100 for this cost)
            net.set_weight_split_levels(2)
            e3, _, _ = measure(k2, min(a.warmup, 10))
            out["two_term_weights"] = {"levels": 2, "dtype": a.dtype, "value": P * k2 / (T_STEPS * e3), "unit": "patches/s",
                                       "steps": k2, "ms_per_step": 1e3 * e3 / k2}
            net.set_weight_split_levels(0)
    if rank == 0 and world == 1 and a.dtype in ("bf16", "fp16") and not a.no_other_dtype:
        pin = lowp_pin(a.dtype, dev)                               # the storage dtype's end-to-end distance (golden G16)
        if pin is not None:
            out["dtype_end_to_end"] = pin
    if rank == 0 and world == 1 and not a.no_legs and not a.no_other_dtype and a.dtype in ("bf16", "fp16") and P == 8 and H == 256:
        # The other BASELINE configs' throughput where the driver sees it, AFTER the headline measurement and
        # its per-launch legs: short runs, compact blocks (the whole lines: `--patches 64`, `--workload cfg5`).
        #   cfg4_share -- configs[3]'s share of one GPU: 64 local patches per GPU (two sub-batches of 32), 20 timed steps;
        #   cfg5       -- configs[4] as stated on one GPU: 1x512x512, fp16, DDIM 50 of 1000, OOD + IND branches, fusion: three samples.
        k64 = 20
        e64, jp64, z64 = measure(k64, 5, Pn=64)
        v64 = 64 * k64 / (T_STEPS * e64)
        leg = {"metric": out["metric"], "value": v64, "unit": "patches/s", "steps": k64, "ms_per_step": 1e3 * e64 / k64, "dtype": a.dtype,
               "patches_per_gpu": 64}
        if not a.no_roofline:
            leg["roofline"] = roofline_leg(gd, jp64, tp_value=v64, dtype=a.dtype, H=H, lo=lo, hi=hi, z=z64, nsteps=2, traffic_tag="p64")
        out["cfg4_share"] = dict(_compact_leg(leg), patches_per_gpu=64,
                                 workload="cfg4's per-GPU share: 64 local patches of 3x256x256 per GPU as two concurrent sub-batches of 32")
        del jp64, z64                                          # (the 64-patch plans stay cached in the model: 2 GB of 288)
        out["cfg5"] = _compact_leg(cfg5_run("fp16", 150, 0, 1, dev, None, want_roofline=not a.no_roofline))
    if rank == 0 and world == 1 and not a.no_cpu_baseline:        # contract: CPU baseline on rank 0 at N=1 only
        out["cpu_baseline"] = cpu_baseline(net.cfg, sd, H, P)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
