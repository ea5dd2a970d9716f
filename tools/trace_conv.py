"""GPU box: cycle stamps of one mid-launch workgroup of the generic conv3x3 kernel (LD_CONV_DEBUG=64)."""
import os, sys, ctypes as C
os.environ["LD_CONV_DEBUG"] = "64"
os.environ.setdefault("LD_CONV_NO_C32", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import hip_helpers as hh
from localdiffusion_hallucination_amd import _cabi as cabi

NAMES = ["start", "setup", "loads issued", "coef built", "barrier0", "lds written (wait+transform)", "barrier1",
         "mfma chunk0", "all chunks", "stores issued", "stats + stores drained"]


def run(B, cin, cout, H, W, dtype="bf16", prologue=False, stats=True):
    x = torch.randn(B, H, W, cin, device="cuda").to(hh.TDT[dtype])
    w = hh.pack(torch.randn(cout, cin, 3, 3) * 0.05, dtype, 3)
    b = torch.zeros(cout, device="cuda")
    st = hh.stats_buffer(B, 8) if stats else None
    if prologue:
        gn = (hh.stats_striped(x.float().permute(0, 3, 1, 2), 8), torch.ones(cin, device="cuda"), torch.zeros(cin, device="cuda"), 8)
        src = hh.make_src(x, cin, gn=gn, act=1)
    else:
        src = hh.make_src(x, cin)
    for _ in range(5):
        hh.conv3x3([src], w, b, B, H, W, cout, dtype, stats=st)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 24)()
    fn = cabi.lib().ld_debug_conv_trace
    fn.restype, fn.argtypes = C.c_int, [C.POINTER(C.c_ulonglong)]
    assert fn(buf) == 0
    t = [buf[k] for k in range(24)]
    print(f"== {cin}->{cout}@{H}x{W} B{B} prologue={prologue} stats={stats}")
    prev = t[0]
    for k, n in enumerate(NAMES):
        if t[k]:
            print(f"   {n:32s} +{t[k] - prev:6d}   (at {t[k] - t[0]})")
            prev = t[k]
    if t[16] and t[19]:      # the statistics epilogue
        print(f"   statistics: barrier +{t[16] - t[9]}, 16-lane sums + LDS +{t[17] - t[16]}, barrier +{t[18] - t[17]}, "
              f"fp64 group sums + atomics +{t[19] - t[18]}, store drain +{t[10] - t[19]}")
    if t[11] and t[15]:      # steady state: chunk 2 of the loop
        for k, n in ((12, "chunk 2: LDS written (vmcnt wait + transform + writes)"), (13, "chunk 2: barrier"),
                     (14, "chunk 2: next chunk's loads issued"), (15, "chunk 2: fragment reads + MFMAs")):
            print(f"   {n:56s} +{t[k] - t[k - 1]:6d}")
        print(f"   chunk 2 total {t[15] - t[11]}; the barrier that opened it was reached at {t[11] - t[0]}")
        print("   (caveat: in the traced build hipcc puts an s_waitcnt vmcnt(0) behind the last, partial weight load of a "
              "chunk -- the stamp that follows copies a register -- so 'loads issued' contains the loads' LATENCY; the "
              "production kernel has no such wait and pays it at the next chunk's LDS write instead)")


if __name__ == "__main__":
    if os.environ.get("LD_TRACE_SHAPES"):          # "B,cin,cout,H,W,prologue;..."
        for t in os.environ["LD_TRACE_SHAPES"].split(";"):
            v = [int(x) for x in t.split(",")]
            run(v[0], v[1], v[2], v[3], v[4], prologue=bool(v[5]))
        sys.exit(0)
    run(8, 32, 32, 256, 256, stats=True)
    run(8, 32, 32, 256, 256, prologue=True)
    run(8, 64, 32, 256, 256, stats=True)
    run(8, 64, 64, 128, 128, prologue=True)
    run(8, 256, 256, 32, 32, prologue=True)
