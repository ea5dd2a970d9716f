"""GPU box: per-phase cycle stamps of a stage program (csrc/stage.hip) next to the stand-alone launches' durations.
usage: python tools/trace_stage.py [B] [H]       (cfg3's denoiser, bf16; LD_STAGE_MAX_PX=1024 is set here)"""
import os, sys, ctypes as C
os.environ.setdefault("LD_STAGE_MAX_PX", "1024")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import localdiffusion_hallucination_amd as ldh
from localdiffusion_hallucination_amd import _cabi as cabi, rng, weights

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
H = int(sys.argv[2]) if len(sys.argv) > 2 else 256
MAXPH = 48
for staged in (False, True):
    net = ldh.Unet(dim=32, init_dim=32, channels=3, out_dim=3, mode="mvtec", compute_dtype="bf16")
    net.load_state_dict({k: torch.from_numpy(v) for k, v in weights.procedural_state_dict(net.cfg, 0).items()})
    net = net.to("cuda")
    net.stage_max_px = 1024 if staged else 0
    p = net.plan(B, H, H, table_T=1000)
    p.x_in.copy_(torch.from_numpy(rng.randn((B, 3, H, H), 1, 100)))
    p.cond_in.copy_(torch.from_numpy(rng.uniform((B, 3, H, H), 1, 101, 0.0, 2.0)))
    st = torch.cuda.current_stream().cuda_stream
    p.run_cond(st)
    p.set_step(500)
    for _ in range(3):
        p.run_main(st)
    acc = {}
    for _ in range(5):
        p.run_main_timed(st, acc)
    torch.cuda.synchronize()
    print(f"== {'stage programs' if staged else 'stand-alone launches'}: B={B} {H}x{H}, {len(p.ops_main)} launches, kernel sum {sum(v[0] for v in acc.values()) / 5 * 1e3:.1f} us")
    for i in sorted(acc):
        m = p.meta.get(i, {})
        if staged and m.get("family") == "stage":
            print(f"   op {i:3d} {m['what']:22s} {m.get('shape', ''):22s} {acc[i][0] / acc[i][1] * 1e3:7.1f} us   fused: {' '.join(m['fused'])}")
        elif not staged and ("32x32" in m.get("shape", "") or m.get("family") == "gn_apply"):
            print(f"   op {i:3d} {m.get('family', '?'):22s} {m.get('shape', ''):22s} {acc[i][0] / acc[i][1] * 1e3:7.1f} us")
    if staged:
        buf = (C.c_ulonglong * (MAXPH * 5 + 2))()
        fn = cabi.lib().ld_debug_stage_trace
        fn.restype, fn.argtypes = C.c_int, [C.POINTER(C.c_ulonglong), C.c_int]
        assert fn(buf, MAXPH * 5 + 2) == 0
        n, t0 = int(buf[MAXPH * 5 + 1]), buf[MAXPH * 5]
        print(f"   last stage launch ({n} phases), cycles of one workgroup: phase | start | wait for 1st tile | tiles | drain | boundary | total")
        for ph in range(n):
            s0, s1, s2, s3, s4 = (buf[ph * 5 + k] for k in range(5))
            first = (s1 - s0) if s1 else 0
            print(f"     {ph:2d} | at {s0 - t0:7d} | {first:6d} | {s2 - (s1 or s0):7d} | {s3 - s2:6d} | {s4 - s3:6d} | {s4 - s0:7d}")
