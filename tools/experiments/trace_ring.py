"""GPU box: correctness (against F.conv2d and, bit for bit, against the generic kernel), launch times and cycle stamps of
the LDS-DMA ring small-map convolution (conv3x3_ring.hip).
usage: LD_LIB_OVERRIDE=<library built with csrc/build.sh --debug-variants> python tools/experiments/trace_ring.py"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.nn.functional as F
import hip_helpers as hh
from localdiffusion_hallucination_amd import _cabi as cabi

NAMES = ["start", "addresses + bias requested", "first chunk(s) requested", "chunk 0 landed + barrier", "ch2: own DMAs landed",
         "ch2: barrier", "ch2: next chunk requested", "ch2: MFMAs", "loop end", "stores issued", "statistics", "stores drained"]
lib = cabi.lib()
lib.ld_debug_ring_enable.restype, lib.ld_debug_ring_enable.argtypes = C.c_int, [C.c_int]


def run(on, srcs, w, b, B, H, W, cout, dtype, stats, add=None):
    lib.ld_debug_ring_enable(on)
    n0 = lib.ld_counter(cabi.COUNTER_CONV3X3_GENERIC)
    if stats is not None:
        stats.zero_()
    out = hh.conv3x3(srcs, w, b, B, H, W, cout, dtype, stats=stats, groups=8)
    torch.cuda.synchronize()
    return out, 1 - (lib.ld_counter(cabi.COUNTER_CONV3X3_GENERIC) - n0)


def check(B, cins, cout, H, W, dtype="bf16", ups=False):
    xs = [hh.rand((B, c, H // 2 if ups else H, W // 2 if ups else W), 1 + i).to(hh.TDT[dtype]).float() for i, c in enumerate(cins)]
    cin = sum(cins)
    w = hh.rand((cout, cin, 3, 3), 7, -0.05, 0.05).to(hh.TDT[dtype]).float()
    b = hh.rand((cout,), 3)
    xin = torch.cat([F.interpolate(x, scale_factor=2, mode="nearest") if ups else x for x in xs], 1)
    ref = F.conv2d(xin, w, b, padding=1)
    srcs = [hh.make_src(hh.nhwc(x, dtype), c, ups=1 if ups else 0) for x, c in zip(xs, cins)]
    wp = hh.pack(w, dtype, 3)
    st1, st0 = hh.stats_buffer(B, 8), hh.stats_buffer(B, 8)
    o1, took = run(1, srcs, wp, b.to(hh.DEV), B, H, W, cout, dtype, st1)
    o0, _ = run(0, srcs, wp, b.to(hh.DEV), B, H, W, cout, dtype, st0)
    err = hh.rel_err(hh.nchw(o1), ref)
    serr = hh.rel_err(st1.sum(1).cpu(), hh.gn_stats_ref(ref, 8))
    same = torch.equal(o1, o0)
    print(f"{'+'.join(map(str, cins))}->{cout}@{H}x{W} B{B} {dtype} ups={int(ups)}: ring launches {took}  rel err {err:.3e}  stats err {serr:.3e}  "
          f"bit-equal to the generic kernel: {same}")


def bench(B, cin, cout, H, W, dtype="bf16", stats=True, reps=50):
    x = torch.randn(B, H, W, cin, device="cuda").to(hh.TDT[dtype])
    w = hh.pack(torch.randn(cout, cin, 3, 3) * 0.05, dtype, 3)
    b = torch.zeros(cout, device="cuda")
    st = hh.stats_buffer(B, 8) if stats else None
    src = hh.make_src(x, cin)
    res = []
    for on in (0, 1):
        lib.ld_debug_ring_enable(on)
        for _ in range(5):
            hh.conv3x3([src], w, b, B, H, W, cout, dtype, stats=st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            hh.conv3x3([src], w, b, B, H, W, cout, dtype, stats=st)
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 1e3 / reps)
    print(f"{cin}->{cout}@{H}x{W} B{B} {dtype} stats={int(stats)}: generic {res[0]:6.1f} us   ring {res[1]:6.1f} us")


def trace(B, cin, cout, H, W, dtype="bf16", stats=True, reps=20):
    x = torch.randn(B, H, W, cin, device="cuda").to(hh.TDT[dtype])
    w = hh.pack(torch.randn(cout, cin, 3, 3) * 0.05, dtype, 3)
    b = torch.zeros(cout, device="cuda")
    st = hh.stats_buffer(B, 8) if stats else None
    src = hh.make_src(x, cin)
    lib.ld_debug_ring_enable(1)
    for _ in range(reps):
        hh.conv3x3([src], w, b, B, H, W, cout, dtype, stats=st)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 16)()
    fn = lib.ld_debug_ring_trace
    fn.restype, fn.argtypes = C.c_int, [C.POINTER(C.c_ulonglong)]
    assert fn(buf) == 0
    t = [buf[k] for k in range(16)]
    print(f"== trace {cin}->{cout}@{H}x{W} B{B} stats={stats}")
    prev = t[0]
    for k, n in enumerate(NAMES):
        if t[k]:
            print(f"   {n:36s} +{t[k] - prev:6d}   (at {t[k] - t[0]})")
            prev = t[k]


if __name__ == "__main__":
    if os.environ.get("LD_CONV_RING_TRACE"):
        trace(4, 256, 256, 32, 32)
        trace(8, 256, 256, 32, 32)
        trace(4, 256, 256, 32, 32, stats=False)
        sys.exit(0)
    for shp in [(2, [256], 256, 32, 32), (3, [128], 128, 32, 32), (1, [64], 64, 64, 64), (2, [64], 128, 24, 48), (2, [128], 256, 16, 32),
                (1, [256], 32, 8, 16), (2, [256, 128], 256, 32, 32), (2, [128, 64], 128, 64, 64)]:
        for dt in ("bf16", "fp16"):
            check(*shp, dtype=dt)
    check(2, [256], 128, 64, 64, ups=True)
    check(2, [128], 64, 32, 32, dtype="fp16", ups=True)
    for shp in [(4, 256, 256, 32, 32), (8, 256, 256, 32, 32), (4, 128, 128, 32, 32), (4, 128, 256, 32, 32), (4, 384, 256, 32, 32),
                (4, 64, 64, 64, 64), (4, 128, 128, 64, 64), (4, 256, 128, 64, 64), (8, 128, 128, 64, 64)]:
        bench(*shp)
        bench(*shp, stats=False)
