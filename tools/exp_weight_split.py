"""GPU box: two-term convolution weights (Unet.weight_split_levels) -- what they buy along the cfg2 chain (128^2, T = 1000,
against the reference golden G5) and what they cost per reverse step at cfg3.  usage: python tools/exp_weight_split.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import localdiffusion_hallucination_amd as ldh
ldh.configure_runtime()
from localdiffusion_hallucination_amd import rng, weights
from test_hip_sampler import make

g = np.load(os.path.join(ROOT, "tests", "golden", "g5_cfg2_mri128.npz"))
cond = torch.from_numpy(rng.uniform((1, 1, 128, 128), 5, 1, 0.0, 2.0))
print("cfg2 chain vs the reference golden: max-abs / mean-abs after t = 999, 500, 100, 10, 0")
for dtype in ("bf16", "fp16"):
    for levels in (0, 1, 2, 4):
        gd = make(dict(mode="mri"), dict(data="mri"), 128, 1000, dtype=dtype)
        gd.model.set_weight_split_levels(levels)
        hist = gd.sample(cond.cuda(), None, batch_size=1, min_max_val=(0.0, 2.0), return_all_timesteps=True).cpu().numpy()
        cells = []
        for t in (999, 500, 100, 10, 0):
            d = np.abs(hist[:, 1000 - t] - g[f"x_after_t{t}"])
            cells.append(f"t={t}: {d.max():.2e} / {d.mean():.2e}")
        print(f"  {dtype} two-term weights on {levels} levels:  " + "  ".join(cells), flush=True)

# cost: cfg3 reverse steps (8 patches of 3x256x256, two sub-batches), as bench.py times them
print("cfg3 step time (8 patches, two concurrent sub-batches, 300 steps after 30):")
for levels in (0, 1, 2, 4, 0):
    net = ldh.Unet(dim=32, init_dim=32, channels=3, out_dim=3, mode="mvtec", compute_dtype="bf16")
    net.load_state_dict({k: torch.from_numpy(v) for k, v in weights.procedural_state_dict(net.cfg, 0).items()})
    net = net.to("cuda")
    net.set_weight_split_levels(levels)
    cfg = dict(branch_out=False, start_intermediate=False, start_timestep=2, data="mvtec", mask_x=False, ood_AD=False,
               ood_confidence=False, classifier=False, use_gt=False)
    gd = ldh.GaussianDiffusion(cfg, net, image_size=256, timesteps=1000, objective="pred_x0", beta_schedule="sigmoid").to("cuda")
    gd.noise_source = "device"
    jp = net.plan(8, 256, 256, table_T=1000)
    jp.cond_in.copy_(torch.from_numpy(rng.uniform((8, 3, 256, 256), 3, 1, 0.0, 2.0)))
    x = torch.empty(8, 3, 256, 256, device="cuda")
    gd._noise(x, 0)
    jp.x_in.copy_(x)
    z = torch.empty_like(x)
    gd.encode_cond(jp, 30)
    draw = gd.run_joint_steps(jp, 999, 30, 0.0, 2.0, z, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gd.run_joint_steps(jp, 969, 300, 0.0, 2.0, z, draw)
    torch.cuda.synchronize()
    print(f"  two-term weights on {levels} levels: {(time.perf_counter() - t0) / 300 * 1e3:.4f} ms per step", flush=True)
