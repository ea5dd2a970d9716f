"""GPU box: step time of two concurrent sub-batches as a function of their relative phase (DESIGN finding 45).
Sub-batch 1 starts d ms after sub-batch 0 (a spin kernel on its stream); K steps each; reported: (wall - d) / K for
small K (the phase holds) and the time each stream needs for its K steps.  usage: python tools/exp_phase.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import localdiffusion_hallucination_amd as ldh
from localdiffusion_hallucination_amd import weights, _cabi as cabi

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
gd.encode_cond(jp, 10)
draw = gd.run_joint_steps(jp, T - 1, 10, 0.0, 2.0, z, 1)
torch.cuda.synchronize()
sub = gd._subs[(id(jp), gd.sub_batches)]
lib = cabi.lib()
g = [v for k, v in sorted(sub.graphs.items(), key=lambda kv: kv[0][0])]
s0, s1 = sub.streams

# calibrate torch.cuda._sleep
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with torch.cuda.stream(s1):
    e0.record(); torch.cuda._sleep(10_000_000); e1.record()
torch.cuda.synchronize()
per_ms = 10_000_000 / e0.elapsed_time(e1)
print(f"_sleep: {per_ms:.0f} cycles per ms")

def run(d_ms, K):
    for sp in sub.plans: sp.set_step(500)
    torch.cuda.synchronize()
    a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    with torch.cuda.stream(s1):
        if d_ms > 0: torch.cuda._sleep(int(d_ms * per_ms))
        b0.record()
    with torch.cuda.stream(s0):
        a0.record()
    for k in range(K):
        cabi.check(lib.ld_graph_launch(g[0], s0.cuda_stream), "l")
        cabi.check(lib.ld_graph_launch(g[1], s1.cuda_stream), "l")
    with torch.cuda.stream(s0): a1.record()
    with torch.cuda.stream(s1): b1.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
    return wall, a0.elapsed_time(a1), b0.elapsed_time(b1)

for K in (2, 4, 8):
    for d in (0.0, 0.2, 0.4, 0.6, 0.8, 1.0, 1.2, 1.4):
        best = min((run(d, K) for _ in range(5)), key=lambda r: r[0])
        print(f"K={K} offset {d:3.1f} ms: wall {best[0]:7.3f}  (wall - offset) / K = {(best[0] - d) / K:6.3f} ms; "
              f"stream 0: {best[1] / K:6.3f} ms per step, stream 1: {best[2] / K:6.3f}")
