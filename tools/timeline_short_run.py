"""GPU box: timeline of a short timed region (the driver's --steps 20 --warmup 5) of the two-sub-batch regime: where the
per-sample fixed part goes (docs/findings.md 50, 64).  Events on each stream after the scatter, the encoder and steps
0, 1, 2, K-1; host clock at the same points.  usage: python tools/timeline_short_run.py [K]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import localdiffusion_hallucination_amd as ldh                                   # noqa: E402

ldh.configure_runtime()
import torch                                                                      # noqa: E402
from localdiffusion_hallucination_amd import _cabi as cabi, diffusion as dm, weights   # noqa: E402

dev = torch.device("cuda:0")
H, P, T, W = 256, 8, 1000, 5
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
net = ldh.Unet(dim=32, init_dim=32, channels=3, out_dim=3, mode="mvtec", compute_dtype="bf16")
net.load_state_dict({k: torch.from_numpy(v) for k, v in weights.procedural_state_dict(net.cfg, 0).items()})
net = net.to(dev)
config = dict(branch_out=False, start_intermediate=False, start_timestep=2, data="mvtec", mask_x=False,
              ood_AD=False, ood_confidence=False, classifier=False, use_gt=False)
gd = ldh.GaussianDiffusion(config, net, image_size=H, timesteps=T, objective="pred_x0", beta_schedule="sigmoid").to(dev)
gd.noise_source = "device"
tn = gd.tuning
jp = net.plan(P, H, H, table_T=T)
jp.cond_in.uniform_(0.0, 2.0)
jp.x_in.normal_()
z = torch.empty_like(jp.x_in)
lib = cabi.lib()


def ev():
    return torch.cuda.Event(enable_timing=True)


for rep in range(5):
    gd.encode_cond(jp, W)
    draw = gd.run_joint_steps(jp, T - 1, W, 0.0, 2.0, z, 1)
    torch.cuda.synchronize()
    sub = gd._subs[(id(jp), gd.sub_batches)]
    cur = torch.cuda.current_stream()
    t_start = T - 1 - W
    base = draw + t_start
    marks, host = {}, {}
    e0 = ev()
    e0.record(cur)
    h0 = time.perf_counter()
    for i, (sp, gs) in enumerate(zip(sub.plans, sub.streams)):
        gs.wait_stream(cur)
        with torch.cuda.stream(gs):
            sp.x_in.copy_(jp.x_in[i * sub.b:(i + 1) * sub.b])
            sp.cond_in.copy_(jp.cond_in[i * sub.b:(i + 1) * sub.b])
            marks[(i, "copy")] = ev()
            marks[(i, "copy")].record(gs)
            sp.run_cond_replayed(gs)
            sp.set_step(t_start + 1)
            marks[(i, "enc")] = ev()
            marks[(i, "enc")].record(gs)
        host[("enc", i)] = time.perf_counter() - h0
    ex = [sub.graphs[(i, 0.0, 2.0, base, gd.noise_seed, gd.noise_offset)] for i in range(sub.S)]
    pace = dm._Pace(sub.streams, tn.sub_ahead)
    for k in range(K):
        if tn.sub_resync > 0 and (k % tn.sub_resync == 0 or k <= tn.sub_resync_early):
            dm._align_streams(sub.streams)
        for i, gs in enumerate(sub.streams):
            cabi.check(lib.ld_graph_launch(ex[i], gs.cuda_stream), "l")
            if k in (0, 1, 2, K - 1) or k % 10 == 0:
                marks[(i, k)] = ev()
                marks[(i, k)].record(gs)
        pace.step_enqueued()
        if k in (0, 1, 2):
            host[("step", k)] = time.perf_counter() - h0
    host["launched"] = time.perf_counter() - h0
    for i, (sp, gs) in enumerate(zip(sub.plans, sub.streams)):
        with torch.cuda.stream(gs):
            jp.x_in[i * sub.b:(i + 1) * sub.b].copy_(sp.x_in)
        cur.wait_stream(gs)
    e1 = ev()
    e1.record(cur)
    host["all"] = time.perf_counter() - h0
    torch.cuda.synchronize()
    host["synced"] = time.perf_counter() - h0
    draw += K
    if rep == 0:
        continue
    tot = e0.elapsed_time(e1)
    print(f"rep {rep}: K={K}: GPU total {tot:.3f} ms = {tot / K:.4f} per step; host wall to sync {1e3 * host['synced']:.3f} ms = "
          f"{1e3 * host['synced'] / K:.4f} per step; host: encoders enqueued at {1e3 * host[('enc', 0)]:.2f} / {1e3 * host[('enc', 1)]:.2f} ms, "
          f"steps 0 / 1 / 2 enqueued at {1e3 * host[('step', 0)]:.2f} / {1e3 * host[('step', 1)]:.2f} / {1e3 * host[('step', 2)]:.2f}, "
          f"last at {1e3 * host['launched']:.2f}")
    for i in range(sub.S):
        m = lambda k: e0.elapsed_time(marks[(i, k)])           # noqa: E731
        print(f"   stream {i}: scatter done {m('copy'):.3f}, encoder done {m('enc'):.3f}, step 0 done {m(0):.3f}, step 1 {m(1):.3f}, "
              f"step 2 {m(2):.3f}, step {K - 1} {m(K - 1):.3f}; steady step {(m(K - 1) - m(2)) / (K - 3):.4f}")
        if K > 30:            # how the step time settles: mean step time per window of 10 steps
            ks = [k for k in range(10, K, 10)]
            print("      per 10 steps from step 10 on: " + " ".join(f"{(m(b_) - m(a_)) / 10:.4f}" for a_, b_ in zip(ks[:-1], ks[1:])))
