"""GPU box: the per-sample fixed part through the library calls bench.py uses (encode_cond + run_joint_steps) for K = 20 / 400 / 20 /
100 / 20 timed steps in ONE process: long-run step vs short runs, first (cold) short run vs later ones (docs/findings.md 86)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import localdiffusion_hallucination_amd as ldh
ldh.configure_runtime()
import torch
from localdiffusion_hallucination_amd import weights
dev = torch.device("cuda:0")
H, P, T, W = 256, 8, 1000, 5
net = ldh.Unet(dim=32, init_dim=32, channels=3, out_dim=3, mode="mvtec", compute_dtype="bf16")
net.load_state_dict({k: torch.from_numpy(v) for k, v in weights.procedural_state_dict(net.cfg, 0).items()})
net = net.to(dev)
config = dict(branch_out=False, start_intermediate=False, start_timestep=2, data="mvtec", mask_x=False,
              ood_AD=False, ood_confidence=False, classifier=False, use_gt=False)
gd = ldh.GaussianDiffusion(config, net, image_size=H, timesteps=T, objective="pred_x0", beta_schedule="sigmoid").to(dev)
gd.noise_source = "device"
jp = net.plan(P, H, H, table_T=T)
jp.cond_in.uniform_(0.0, 2.0); jp.x_in.normal_()
z = torch.empty_like(jp.x_in)
heat = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
for it, K in enumerate((20, 20, 400, 20, 100, 20)):
    gd.encode_cond(jp, W)
    draw = gd.run_joint_steps(jp, T - 1, W, 0.0, 2.0, z, 1)
    if it == 1 and os.environ.get("LD_PREHEAT"):        # the second cold-ish run behind ~100 ms of matrix products: clock ramp or not?
        for _ in range(int(os.environ["LD_PREHEAT"])):
            heat @ heat
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gd.encode_cond(jp, K)
    t1 = time.perf_counter()
    draw = gd.run_joint_steps(jp, T - 1 - W, K, 0.0, 2.0, z, draw)
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print(f"K={K}: total {1e3*(t3-t0):.3f} ms = {1e3*(t3-t0)/K:.4f}/step; host: encode_cond {1e3*(t1-t0):.3f}, run_joint_steps returns at {1e3*(t2-t0):.3f}, synced {1e3*(t3-t0):.3f}")
