"""Experiment: two sub-batches of 4 whose small-map sections (64^2, 32^2: latency-bound) are forced to alternate, so that
one sub-batch's small-map section always runs beside the other's large-map section (256^2, 128^2: HBM-bound).
Each sub-batch's step is cut into two graphs: SMALL = ops[lo:hi], BIG = ops[hi:] + final step + counter + the next
step's memsets + ops[:lo].  Usage: python tools/exp_phased.py [steps] [lo] [hi]"""
import ctypes as C
import sys
import time

import torch

sys.path.insert(0, ".")
import localdiffusion_hallucination_amd as ldh                          # noqa: E402
from localdiffusion_hallucination_amd import _cabi as cabi, weights     # noqa: E402

T, H = 1000, 256
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
LO = int(sys.argv[2]) if len(sys.argv) > 2 else 20
HI = int(sys.argv[3]) if len(sys.argv) > 3 else 79
dev = torch.device("cuda", 0)
net = ldh.Unet(dim=32, init_dim=32, channels=3, out_dim=3, mode="mvtec", compute_dtype="bf16")
net.load_state_dict({k: torch.from_numpy(v) for k, v in weights.procedural_state_dict(net.cfg, 0).items()})
net = net.to(dev)
config = dict(branch_out=False, start_intermediate=False, start_timestep=2, data="mvtec", mask_x=False,
              ood_AD=False, ood_confidence=False, classifier=False, use_gt=False)
gd = ldh.GaussianDiffusion(config, net, image_size=H, timesteps=T, objective="pred_x0", beta_schedule="sigmoid").to(dev)
lib = cabi.lib()
sched = gd._sched_table()


def make(B, inst):
    p = net.plan(B, H, H, table_T=T, instance=inst)
    p.cond_in.uniform_(0.0, 2.0)
    p.x_in.normal_()
    p.run_cond(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return p


def memsets(p, st):
    s = p.stats[p.cond_slots:]
    cabi.check(lib.ld_memset_zero(s.data_ptr(), s.numel() * 8, st), "memset")
    cabi.check(lib.ld_memset_zero(p.kmax_arena.data_ptr(), p.kmax_arena.numel() * 4, st), "memset")


def final(p, i, st):
    xa, wf, bf = p.final
    B_, C_, H_, W_ = p.model_out.shape
    cabi.check(lib.ld_final_step_at(xa.data_ptr(), wf.data_ptr(), bf.data_ptr(), p.model_out.data_ptr(), p.x_in.data_ptr(),
                                    None, sched.data_ptr(), p.t_dev.data_ptr(), 0.0, 2.0, 0, 10, T, -1, i * p.x_in.numel(),
                                    B_, H_, W_, wf.shape[1], C_, p.dt, st), "final_step")
    cabi.check(lib.ld_step_add(p.t_dev.data_ptr(), -1, st), "step_add")


def head(p, st):
    memsets(p, st)
    for op in p.ops_main[:LO]:
        op(st)


def small(p, st):
    for op in p.ops_main[LO:HI]:
        op(st)


def big(p, i, st):
    for op in p.ops_main[HI:-1]:
        op(st)
    final(p, i, st)
    head(p, st)


def capture(fn, gs):
    st = gs.cuda_stream
    cabi.check(lib.ld_graph_begin(st), "begin")
    fn(st)
    ex = C.c_void_p()
    cabi.check(lib.ld_graph_end(st, C.byref(ex)), "end")
    return ex


ps = [make(4, 11), make(4, 12)]
ss = [torch.cuda.Stream(), torch.cuda.Stream()]
gsmall, gbig = [], []
for i, (p, gs) in enumerate(zip(ps, ss)):
    with torch.cuda.stream(gs):
        p.set_step(T - 1)
        head(p, gs.cuda_stream); small(p, gs.cuda_stream); big(p, i, gs.cuda_stream)      # eager once
        gs.synchronize()
        gsmall.append(capture(lambda st: small(p, st), gs))
        gbig.append(capture(lambda st: big(p, i, st), gs))
        gs.synchronize()


def run(phased):
    ev = [[torch.cuda.Event() for _ in range(steps + 1)] for _ in range(2)]
    for p in ps:
        p.set_step(T - 2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    A, B = ss[0], ss[1]
    for k in range(steps):
        if phased and k > 0:
            A.wait_event(ev[1][k - 1])
        cabi.check(lib.ld_graph_launch(gsmall[0], A.cuda_stream), "launch")
        ev[0][k].record(A)
        cabi.check(lib.ld_graph_launch(gbig[0], A.cuda_stream), "launch")
        if phased:
            B.wait_event(ev[0][k])
        cabi.check(lib.ld_graph_launch(gsmall[1], B.cuda_stream), "launch")
        ev[1][k].record(B)
        cabi.check(lib.ld_graph_launch(gbig[1], B.cuda_stream), "launch")
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / steps


for phased in (False, True, False, True):
    print(f"small = ops[{LO}:{HI}], phased={phased}: {run(phased):.3f} ms/step of 8", flush=True)
