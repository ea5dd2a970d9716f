"""GPU box: correctness summary and cycle stamps of the shelved K-split small-map convolution (conv3x3_ksplit.hip).
usage: LD_LIB_OVERRIDE=<library built with csrc/build.sh --debug-variants> python tools/experiments/trace_ksplit.py"""
import os, sys, ctypes as C
os.environ.setdefault("LD_CONV_KSPLIT", "1")
os.environ.setdefault("LD_CONV_KSPLIT_TRACE", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.nn.functional as F
import hip_helpers as hh
from localdiffusion_hallucination_amd import _cabi as cabi

NAMES = ["start", "loads issued (DMA + weights)", "coefficients", "own DMAs landed (+ prologue)", "barrier", "MFMAs",
         "barrier (halo dead)", "partial sums joined", "epilogue stores issued", "statistics", "stores drained"]


def check(B, cin, cout, H, W, dtype="bf16"):
    lib = cabi.lib()
    x = hh.rand((B, cin, H, W), 1).to(hh.TDT[dtype]).float()
    w = hh.rand((cout, cin, 3, 3), 2, -0.05, 0.05).to(hh.TDT[dtype]).float()
    b = hh.rand((cout,), 3)
    ref = F.conv2d(x, w, b, padding=1)
    n0 = lib.ld_counter(cabi.COUNTER_CONV3X3_GENERIC)
    stats = hh.stats_buffer(B, 8)
    out = hh.conv3x3([hh.make_src(hh.nhwc(x, dtype), cin)], hh.pack(w, dtype, 3), b.to(hh.DEV), B, H, W, cout, dtype, stats=stats, groups=8)
    torch.cuda.synchronize()
    took = 1 - (lib.ld_counter(cabi.COUNTER_CONV3X3_GENERIC) - n0)      # the generic kernel counts only what falls through
    err = hh.rel_err(hh.nchw(out), ref)
    serr = hh.rel_err(stats.sum(1).cpu(), hh.gn_stats_ref(ref, 8))
    d = (hh.nchw(out) - ref).abs()
    bad = (d > 0.05 * ref.abs().max()).nonzero()
    print(f"{cin}->{cout}@{H}x{W} B{B} {dtype}: ws launches {took}  rel err {err:.3e}  stats err {serr:.3e}  bad elements {len(bad)}"
          + (f"  first bad (b,c,y,x) {bad[0].tolist()} rows {sorted(set(bad[:, 2].tolist()))[:12]} cols {sorted(set(bad[:, 3].tolist()))[:12]} chans {sorted(set(bad[:, 1].tolist()))[:8]}" if len(bad) else ""))


def trace(B, cin, cout, H, W, dtype="bf16", stats=True, reps=20):
    x = torch.randn(B, H, W, cin, device="cuda").to(hh.TDT[dtype])
    w = hh.pack(torch.randn(cout, cin, 3, 3) * 0.05, dtype, 3)
    b = torch.zeros(cout, device="cuda")
    st = hh.stats_buffer(B, 8) if stats else None
    src = hh.make_src(x, cin)
    for _ in range(reps):
        hh.conv3x3([src], w, b, B, H, W, cout, dtype, stats=st)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 16)()
    try:
        fn = cabi.lib().ld_debug_ksplit_trace
    except AttributeError:
        print("no ld_debug_ksplit_trace in this build")
        return
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
    for shp in [(2, 256, 256, 32, 32), (3, 128, 128, 32, 32), (1, 64, 64, 64, 64), (2, 64, 128, 24, 48), (2, 128, 256, 16, 32), (1, 256, 32, 8, 16)]:
        for dt in ("bf16", "fp16"):
            check(*shp, dtype=dt)
    trace(4, 256, 256, 32, 32)
    trace(8, 256, 256, 32, 32)
    trace(4, 256, 256, 32, 32, stats=False)
