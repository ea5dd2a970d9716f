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
  roofline     -- dominant kernel family: algorithmic bytes / live HIP-event time vs 8 TB/s
  cpu_baseline -- the oracle (plain PyTorch fp32 CPU port of the reference) on this box's cores.
"""
import argparse
import json
import os
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
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--patches", type=int, default=8, help="local patches per GPU (K masks of one image)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--graph", type=int, default=-1, help="1: replay the reverse step from a captured HIP graph (default: eager launches)")
    return ap.parse_args()


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


def cpu_baseline(cfg, sd, H, seconds_budget=15.0):
    """Oracle forward+update on the host cores: bounded sample, fp32, all threads."""
    from oracle import unet_ref
    from localdiffusion_hallucination_amd import rng
    B = 1
    x = torch.from_numpy(rng.randn((B, cfg.channels, H, H), 3, 0))
    cond = torch.from_numpy(rng.uniform((B, cfg.cond_in_channels, H, H), 3, 1, 0.0, 2.0))
    t = torch.full((B,), 500, dtype=torch.long)
    with torch.no_grad():
        # the GPU box exposes far more logical CPUs than this process may use; oversubscribing the
        # intra-op pool makes the oracle ~100x slower, so pick the fastest of a few pool sizes
        best = None
        for nt in sorted({min(os.cpu_count(), k) for k in (8, 16, 32, 64)}):
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
        n, t0 = 0, time.time()
        while True:
            y = unet_ref.unet_forward(sd, cfg, x, cond, t)
            x = 0.9 * x + 0.1 * y.clamp(0, 2)
            n += 1
            if time.time() - t0 > seconds_budget or n >= 40:
                break
        dt = (time.time() - t0) / n
    return dict(value=1.0 / (T_STEPS * dt), unit="patches/s", cores=torch.get_num_threads(), kind="port",
                sample=f"{n} consecutive reverse steps of 1 patch ({cfg.channels}x{H}x{H}, fp32, oracle/unet_ref.py), "
                       f"{dt*1e3:.1f} ms/step, extrapolated to T={T_STEPS}")


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    import localdiffusion_hallucination_amd as ldh
    from localdiffusion_hallucination_amd import _cabi as cabi, rng, weights

    H, P = a.size, a.patches
    net = ldh.Unet(dim=32, init_dim=32, channels=3, out_dim=3, mode="mvtec", compute_dtype=a.dtype)
    sd = {k: torch.from_numpy(v) for k, v in weights.procedural_state_dict(net.cfg, 0).items()}
    net.load_state_dict(sd)
    net = net.to(dev)
    config = dict(branch_out=False, start_intermediate=False, start_timestep=2, data="mvtec", mask_x=False,
                  ood_AD=False, ood_confidence=False, classifier=False, use_gt=False)
    gd = ldh.GaussianDiffusion(config, net, image_size=H, timesteps=T_STEPS, objective="pred_x0",
                               beta_schedule="sigmoid").to(dev)
    gd.noise_source = "device"
    gd.use_graph = a.graph == 1          # HIP-graph replay of the reverse step (measured: no gain, the step is GPU-bound)

    masks = band_masks(P, H)
    cond_img = torch.from_numpy(rng.uniform((1, 3, H, H), 100 + rank, 1, 0.0, 2.0))
    conds = patch_conditions(cond_img, masks).to(dev)
    lib, st = cabi.lib(), torch.cuda.current_stream().cuda_stream
    jp = net.plan(P, H, H, table_T=T_STEPS)
    jp.cond_in.copy_(conds)
    x_T = torch.empty(P, 3, H, H, device=dev)
    gd._noise(x_T, 0)
    jp.x_in.copy_(x_T)
    z = torch.empty_like(x_T)
    lo, hi = 0.0, 2.0

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # warm-up (untimed): encoder + W steps
    jp.run_cond(st)
    draw = gd.run_joint_steps(jp, T_STEPS - 1, a.warmup, lo, hi, z, 1)
    sync_all()
    t0 = time.perf_counter()
    jp.run_cond(st)                                             # once per sample, inside the timed region
    t_start = T_STEPS - 1 - a.warmup
    done = 0
    while done < a.steps:                                       # wrap to a new sample after T steps
        chunk = min(a.steps - done, t_start + 1)
        draw = gd.run_joint_steps(jp, t_start, chunk, lo, hi, z, draw)
        done += chunk
        t_start -= chunk
        if t_start < 0:
            t_start = T_STEPS - 1
    # per-sample exchange: all-gather the local patches and recompose by the masks
    xl = jp.x_in
    if world > 1:
        gathered = torch.empty(world * P, 3, H, H, device=dev)
        dist.all_gather_into_tensor(gathered, xl.contiguous())
        xl = gathered[rank * P:(rank + 1) * P]
    img = torch.empty(1, 3, H, H, device=dev)
    mk = masks.reshape(P, H * H).to(dev)
    cabi.check(lib.ld_recompose(xl.contiguous().data_ptr(), mk.data_ptr(), img.data_ptr(), 1, P, 3, H * H, st), "recompose")
    sync_all()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    assert torch.isfinite(img).all(), "non-finite output"

    value = world * P * a.steps / (T_STEPS * elapsed)
    out = {
        "metric": "local patches/sec (256^2, T=1000)", "value": value, "unit": "patches/s", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype,
        "data": "synthetic (portable-RNG conditioning image and x_T, procedural random-init weights)",
        "config": {"workload": f"cfg3: {P} local patches (vertical band masks) of one 3x{H}x{H} image per GPU, "
                               f"4-stage dim-32 conditional UNet (12.1M params), DDPM T={T_STEPS}, pred_x0, sigmoid schedule",
                   "patches_per_gpu": P, "image": [3, H, H], "timesteps": T_STEPS,
                   "parallelism": f"patch-sharded x{world}, one all-gather per sample",
                   "hip_graph": bool(gd.use_graph),
                   "concurrent_sub_batches": (gd.sub_batches if gd.timed_plan(jp) is not jp else 1)},
    }

    if rank == 0 and not a.no_roofline:
        # dominant kernel family: live HIP events around every launch, on the launch stream
        acc = {}
        torch.cuda.synchronize()
        gd.run_joint_steps(jp, 500, 5, lo, hi, z, 1, timers=acc)
        tp = gd.timed_plan(jp)            # sub-batch 0 (timed beside the other sub-batch) when the step runs split
        fam = {}
        total_ms = 0.0
        for i, (ms, cnt) in acc.items():
            m = tp.meta.get(i, {})
            total_ms += ms
            f = fam.setdefault(m.get("family", m.get("what", "other").split(" ")[0]), dict(ms=0.0, launches=0, bytes=0, flops=0))
            f["ms"] += ms
            f["launches"] += cnt
            f["bytes"] += m.get("bytes", 0) * cnt
            f["flops"] += m.get("flops", 0) * cnt
        if os.environ.get("LD_BENCH_OPS"):
            with open(os.environ["LD_BENCH_OPS"], "w") as f:
                for i in sorted(acc):
                    ms, cnt = acc[i]
                    m = tp.meta.get(i, {})
                    us = 1e3 * ms / cnt
                    f.write(f"{i:4d} {m.get('family', '?'):22s} {m.get('what', '?'):38s} {m.get('shape', ''):20s} "
                            f"{us:9.1f} us  {m.get('bytes', 0) / max(us, 1e-9) / 1e3:8.1f} GB/s  "
                            f"{m.get('flops', 0) / max(us, 1e-9) / 1e6:8.1f} TF/s\n")
        # the north star's "ResBlock conv path": the C=32 3x3 convolutions at full resolution (SURVEY 8a census, first
        # two lines), priced against the HBM roofline with their algorithmic bytes
        rb_ms = rb_bytes = rb_n = 0
        for i, (ms, cnt) in acc.items():
            m = tp.meta.get(i, {})
            if m.get("family", "").startswith("conv3x3") and m.get("shape", "").endswith(f"@{H}x{H}"):
                rb_ms += ms
                rb_bytes += m.get("bytes", 0) * cnt
                rb_n += cnt
        name, d = max(fam.items(), key=lambda kv: kv[1]["ms"])
        sec = d["ms"] * 1e-3
        gbs = d["bytes"] / sec / 1e9 if d["bytes"] else 0.0
        tfs = d["flops"] / sec / 1e12 if d["flops"] else 0.0
        peak_tf = MFMA_PEAK_TF[a.dtype]
        hbm_frac, mfma_frac = gbs / HBM_PEAK_GBS, tfs / peak_tf
        # a kernel is priced against the roofline that bounds it: the larger of the two fractions
        if mfma_frac > hbm_frac:
            roof = {"bound": "mfma", "achieved": tfs, "peak": peak_tf, "unit": "TFLOP/s", "frac": mfma_frac}
        else:
            roof = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_frac}
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")   # committed PMC pass (rocprofv3 --pmc), not live
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get(name, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roof.update({"kernel": name, "launch_batch": int(tp.x_in.shape[0]),
                     # launches of that many independent sub-batches share the GPU while this one is timed
                     "concurrent_streams": (gd.sub_batches if tp is not jp else 1), "traffic": traffic, "avg_launch_us": 1e3 * d["ms"] / max(1, d["launches"]),
                     "algorithmic_GBps": gbs, "algorithmic_TFLOPps": tfs,
                     "share_of_step": d["ms"] / max(total_ms, 1e-9),
                     # whole path: 0.7036 TB of algorithmic traffic per 256^2 bf16 patch over T=1000
                     "path_frac": value / world * ALGO_TB_PER_PATCH * 1e3 / HBM_PEAK_GBS,
                     "resblock_conv_path": {"launches_per_step": rb_n // 5, "ms_per_step": round(rb_ms / 5, 4),
                                            "algorithmic_GBps": rb_bytes / max(rb_ms, 1e-9) / 1e6,
                                            "hbm_frac": rb_bytes / max(rb_ms, 1e-9) / 1e6 / HBM_PEAK_GBS},
                     "families_ms_per_step": {k: round(v["ms"] / 5, 4) for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms"])}})
        out["roofline"] = roof
    if rank == 0 and world == 1 and not a.no_cpu_baseline:        # contract: CPU baseline on rank 0 at N=1 only
        out["cpu_baseline"] = cpu_baseline(net.cfg, sd, H)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
