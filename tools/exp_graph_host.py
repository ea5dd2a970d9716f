"""GPU box: host cost of replaying a sub-batch's step graph (DESIGN finding 29/40).
usage: python tools/exp_graph_host.py [steps]   (env: LD_SUB_BATCHES, DEBUG_CLR_GRAPH_PACKET_CAPTURE, ...)
Prints host microseconds per hipGraphLaunch and per kernel node for a short burst (the hardware queue cannot fill)
and for a long one (back-pressure shows as host time ~ wall time)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import localdiffusion_hallucination_amd as ldh
from localdiffusion_hallucination_amd import weights

def main():
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
    jp.cond_in.normal_()
    x = torch.randn(P, 3, H, H, device=dev)
    jp.x_in.copy_(x)
    z = torch.empty_like(x)
    gd.encode_cond(jp, 50)
    draw = gd.run_joint_steps(jp, T - 1, 50, 0.0, 2.0, z, 1)
    torch.cuda.synchronize()
    sub = gd._subs.get((id(jp), gd.sub_batches))
    nodes = 106
    for steps in (10, 40, int(sys.argv[1]) if len(sys.argv) > 1 else 400):
        sub.host_launch_s, sub.host_launch_n = 0.0, 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        draw = gd.run_joint_steps(jp, 900, steps, 0.0, 2.0, z, draw)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"S={sub.S} steps={steps:4d}: host in launch loop {1e6*sub.host_launch_s/sub.host_launch_n:7.1f} us per graph launch "
              f"(~{1e6*sub.host_launch_s/sub.host_launch_n/nodes:5.2f} us per node at {nodes} nodes); "
              f"host total {1e3*(t1-t0)/steps:6.3f} ms/step, wall {1e3*(t2-t0)/steps:6.3f} ms/step")
main()
