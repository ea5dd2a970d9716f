"""Experiment: does running the 8 patches as TWO independent half-batches on two HIP streams (each a replayed HIP
graph of one reverse step, the second stream started half a step late) beat one batch of 8?  The small-map
section of a step is latency / L2-bound and the large-map section HBM-bound, so the two halves could overlap.
Usage: python tools/exp_two_stream.py [steps]      (prints ms per step of 8 patches for each arrangement)"""
import ctypes as C
import sys
import time

import torch

sys.path.insert(0, ".")
import localdiffusion_hallucination_amd as ldh                          # noqa: E402
from localdiffusion_hallucination_amd import _cabi as cabi, weights     # noqa: E402
from localdiffusion_hallucination_amd.unet import _Plan                 # noqa: E402

T, H = 1000, 256
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda", 0)
net = ldh.Unet(dim=32, init_dim=32, channels=3, out_dim=3, mode="mvtec", compute_dtype="bf16")
net.load_state_dict({k: torch.from_numpy(v) for k, v in weights.procedural_state_dict(net.cfg, 0).items()})
net = net.to(dev)
config = dict(branch_out=False, start_intermediate=False, start_timestep=2, data="mvtec", mask_x=False,
              ood_AD=False, ood_confidence=False, classifier=False, use_gt=False)
gd = ldh.GaussianDiffusion(config, net, image_size=H, timesteps=T, objective="pred_x0", beta_schedule="sigmoid").to(dev)
gd.noise_source = "device"
lib = cabi.lib()
sched = gd._sched_table()
obj = cabi.OBJ["pred_x0"]


def make(B):
    p = _Plan(net, B, H, H, T)
    p.cond_in.uniform_(0.0, 2.0)
    p.x_in.normal_()
    p.run_cond(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return p, torch.empty_like(p.x_in)


def one_step(p, z, st, base):
    p.run_main(st)
    cabi.check(lib.ld_randn(z.data_ptr(), z.numel(), 10, base, -1, p.t_dev.data_ptr(), st), "randn")
    cabi.check(lib.ld_ddpm_step(p.x_in.data_ptr(), p.model_out.data_ptr(), z.data_ptr(), p.x_in.data_ptr(), None,
                                sched.data_ptr(), p.t_dev.data_ptr(), 0.0, 2.0, obj, z.numel(), st), "ddpm_step")
    cabi.check(lib.ld_step_add(p.t_dev.data_ptr(), -1, st), "step_add")


def capture(p, z, gs):
    st = gs.cuda_stream
    with torch.cuda.stream(gs):
        p.set_step(T - 1)
        one_step(p, z, st, T)
        gs.synchronize()
        cabi.check(lib.ld_graph_begin(st), "begin")
        one_step(p, z, st, T)
        ex = C.c_void_p()
        cabi.check(lib.ld_graph_end(st, C.byref(ex)), "end")
    return ex


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / steps


# --- S sub-batches on S streams
def split(S, stagger):
    ps = [make(8 // S) for _ in range(S)]
    ss = [torch.cuda.Stream() for _ in range(S)]
    gs = [capture(p, z, s) for (p, z), s in zip(ps, ss)]

    def f():
        for p, _ in ps:
            p.set_step(T - 2)
        torch.cuda.synchronize()
        if stagger:
            for i, s in enumerate(ss):
                if i:
                    with torch.cuda.stream(s):
                        torch.cuda._sleep(stagger * i)
        for _ in range(steps):
            for g, s in zip(gs, ss):
                cabi.check(lib.ld_graph_launch(g, s.cuda_stream), "launch")
    for _ in range(2):
        print(f"{S} sub-batches of {8 // S}, stagger {stagger}: {timed(f):.3f} ms/step of 8", flush=True)


for S in [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4").split(",")]:
    split(S, 0)
