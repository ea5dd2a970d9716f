"""GPU box: timeline of a short timed region (the driver's --steps 20 --warmup 5) of the two-sub-batch regime: where the
per-sample fixed part goes (DESIGN finding 50).  Events on each stream after the encoder and after steps 0, 1, 2, K-1."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import localdiffusion_hallucination_amd as ldh
from localdiffusion_hallucination_amd import weights, _cabi as cabi, diffusion as dm

dev = torch.device("cuda:0")
H, P, T, W, K = 256, 8, 1000, 5, 20
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
lib = cabi.lib()

def ev():
    return torch.cuda.Event(enable_timing=True)

for rep in range(4):
    gd.encode_cond(jp, W)
    draw = gd.run_joint_steps(jp, T - 1, W, 0.0, 2.0, z, 1)
    torch.cuda.synchronize()
    sub = gd._subs[(id(jp), gd.sub_batches)]
    cur = torch.cuda.current_stream()
    t_start = T - 1 - W
    base = draw + t_start
    marks = {}
    e0 = ev(); e0.record(cur)
    h0 = time.perf_counter()
    host = {}
    for i, (sp, gs) in enumerate(zip(sub.plans, sub.streams)):
        gs.wait_stream(cur)
        with torch.cuda.stream(gs):
            sp.x_in.copy_(jp.x_in[i * sub.b:(i + 1) * sub.b])
            sp.cond_in.copy_(jp.cond_in[i * sub.b:(i + 1) * sub.b])
            sp.run_cond(gs.cuda_stream)
            sp.set_step(t_start + 1)
            marks[(i, "enc")] = ev(); marks[(i, "enc")].record(gs)
        host[("enc", i)] = time.perf_counter() - h0
    ex = [sub.graphs[(i, 0.0, 2.0, base, gd.noise_seed, gd.noise_offset)] for i in range(sub.S)]
    for k in range(K):
        if k % 32 == 0 or k <= dm._RESYNC_EARLY:
            dm._align_streams(sub.streams)
        for i, gs in enumerate(sub.streams):
            cabi.check(lib.ld_graph_launch(ex[i], gs.cuda_stream), "l")
            if k in (0, 1, 2, K - 1):
                marks[(i, k)] = ev(); marks[(i, k)].record(gs)
        if k in (0, 1): host[("step", k)] = time.perf_counter() - h0
    for i, (sp, gs) in enumerate(zip(sub.plans, sub.streams)):
        with torch.cuda.stream(gs):
            jp.x_in[i * sub.b:(i + 1) * sub.b].copy_(sp.x_in)
        cur.wait_stream(gs)
    e1 = ev(); e1.record(cur)
    host["all"] = time.perf_counter() - h0
    torch.cuda.synchronize()
    draw += K
    if rep == 0: continue
    print(f"rep {rep}: total {e0.elapsed_time(e1):.3f} ms = {e0.elapsed_time(e1) / K:.4f} per step; host: encoders enqueued at "
          f"{1e3 * host[('enc', 0)]:.2f} / {1e3 * host[('enc', 1)]:.2f} ms, step 0 at {1e3 * host[('step', 0)]:.2f}, step 1 at {1e3 * host[('step', 1)]:.2f}, all at {1e3 * host['all']:.2f}")
    for i in range(sub.S):
        print(f"   stream {i}: encoder done {e0.elapsed_time(marks[(i, 'enc')]):.3f}, step 0 done {e0.elapsed_time(marks[(i, 0)]):.3f}, step 1 {e0.elapsed_time(marks[(i, 1)]):.3f}, "
              f"step 2 {e0.elapsed_time(marks[(i, 2)]):.3f}, step {K - 1} {e0.elapsed_time(marks[(i, K - 1)]):.3f}")
