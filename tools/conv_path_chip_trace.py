#!/usr/bin/env python3
"""The ResBlock conv path's chip-level figure as its own program, so that a kernel trace can reproduce it.

bench.py's `resblock_conv_path.chip_hbm_frac` is `conv_path_chip_leg`: the twelve C = 32 3x3 launches at 256^2 of BOTH
sub-batches (+ the statistics reset they depend on), one captured graph per stream, replayed concurrently and in phase;
bytes of both sequences / wall.  This script builds the cfg3 sampler exactly as bench.py does, runs that leg alone and prints
its JSON; under the profiler

    rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_cp -o r -- python3 tools/conv_path_chip_trace.py --reps 200
    python3 tools/conv_path_union.py /tmp/prof_cp/.../r_kernel_trace.csv

the trace holds (almost) nothing but those launches, and the union of their execution intervals over both queues is the same
quantity seen from the dispatch timestamps.  tools/profile_round.sh runs both and commits them side by side
(profiles/<tag>_conv_path_chip.txt).  No oracle, no CPU path: product code only."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--patches", type=int, default=8)
    a = ap.parse_args()
    import bench
    import localdiffusion_hallucination_amd as ldh
    from localdiffusion_hallucination_amd import rng, weights
    ldh.configure_runtime()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    H, P, T = 256, a.patches, bench.T_STEPS
    net = ldh.Unet(dim=32, init_dim=32, channels=3, out_dim=3, mode="mvtec", compute_dtype=a.dtype)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in weights.procedural_state_dict(net.cfg, 0).items()})
    net = net.to(dev)
    config = dict(branch_out=False, start_intermediate=False, start_timestep=2, data="mvtec", mask_x=False,
                  ood_AD=False, ood_confidence=False, classifier=False, use_gt=False)
    gd = ldh.GaussianDiffusion(config, net, image_size=H, timesteps=T, objective="pred_x0", beta_schedule="sigmoid").to(dev)
    gd.noise_source = "device"
    masks = bench.band_masks(P, H)
    cond_img = torch.from_numpy(rng.uniform((1, 3, H, H), 100, 1, 0.0, 2.0))
    jp = net.plan(P, H, H, table_T=T)
    jp.cond_in.copy_(bench.patch_conditions(cond_img, masks).to(dev))
    x_T = torch.empty(P, 3, H, H, device=dev)
    gd._noise(x_T, 0)
    jp.x_in.copy_(x_T)
    z = torch.empty_like(x_T)
    gd.encode_cond(jp, 8)
    gd.run_joint_steps(jp, T - 1, 8, 0.0, 2.0, z, 1)           # builds the sub-batch runner (plans, streams, step graphs)
    torch.cuda.synchronize()
    sub = gd._subs.get((id(jp), gd.sub_batches))
    assert sub is not None, "the batch was not split into sub-batches"

    def conv_sel(m):
        return m.get("family", "").startswith("conv3x3") and m.get("shape", "").endswith(f"@{H}x{H}")
    res = bench.conv_path_chip_leg(gd, sub, conv_sel, reps=a.reps)
    res["reps"] = a.reps
    res["launches_per_replay_per_stream"] = len([i for i in sub.plans[0].meta if conv_sel(sub.plans[0].meta[i])])
    print(json.dumps(res))


if __name__ == "__main__":
    main()
