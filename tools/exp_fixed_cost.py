"""GPU box: the per-sample fixed part of bench.py's timed region (DESIGN finding 44): K timed steps for several K,
with and without the conditioning encoder, wall and host-enqueue time.  usage: python tools/exp_fixed_cost.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import localdiffusion_hallucination_amd as ldh
from localdiffusion_hallucination_amd import weights, rng

dev = torch.device("cuda:0")
H, P, T = 256, 8, 1000
net = ldh.Unet(dim=32, init_dim=32, channels=3, out_dim=3, mode="mvtec", compute_dtype="bf16")
net.load_state_dict({k: torch.from_numpy(v) for k, v in weights.procedural_state_dict(net.cfg, 0).items()})
net = net.to(dev)
config = dict(branch_out=False, start_intermediate=False, start_timestep=2, data="mvtec", mask_x=False,
              ood_AD=False, ood_confidence=False, classifier=False, use_gt=False)
gd = ldh.GaussianDiffusion(config, net, image_size=H, timesteps=T, objective="pred_x0", beta_schedule="sigmoid").to(dev)
gd.noise_source = "device"
jp = net.plan(P, H, H, table_T=T)
jp.cond_in.uniform_(0.0, 2.0)
jp.x_in.normal_()
z = torch.empty_like(jp.x_in)
W = 5
for encode in (True, False):
    for K in (1, 2, 5, 10, 20, 50, 100):
        best = None
        for rep in range(5):
            gd.encode_cond(jp, W)
            draw = gd.run_joint_steps(jp, T - 1, W, 0.0, 2.0, z, 1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if encode:
                gd.encode_cond(jp, K)
            draw = gd.run_joint_steps(jp, T - 1 - W, K, 0.0, 2.0, z, draw)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            r = (t2 - t0, t1 - t0)
            best = r if best is None or r[0] < best[0] else best
        print(f"encoder {'in ' if encode else 'out'} K={K:4d}: wall {1e3 * best[0]:7.3f} ms ({1e3 * best[0] / K:6.3f} per step), host enqueue {1e3 * best[1]:6.3f} ms")
