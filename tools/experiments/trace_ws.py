"""GPU box: phase trace of the wave-specialised conv (LD_CONV_DEBUG=32): where one workgroup's time goes."""
import os, sys, ctypes as C
os.environ["LD_CONV_DEBUG"] = str(32 | int(os.environ.get("LD_TRACE_EXTRA", "0")))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import hip_helpers as hh
from localdiffusion_hallucination_amd import _cabi as cabi

def run(B, cin, cout, H, W, dtype="bf16", prologue=False):
    x = torch.randn(B, H, W, cin, device="cuda").to(hh.TDT[dtype])
    w = hh.pack(torch.randn(cout, cin, 3, 3) * 0.05, dtype, 3)
    b = torch.zeros(cout, device="cuda")
    st = hh.stats_buffer(B, 8)
    if prologue:
        gn = (hh.stats_striped(x.float().permute(0, 3, 1, 2), 8), torch.ones(cin, device="cuda"), torch.zeros(cin, device="cuda"), 8)
        src = hh.make_src(x, cin, gn=gn, act=1)
    else:
        src = hh.make_src(x, cin)
    for _ in range(5):
        hh.conv3x3([src], w, b, B, H, W, cout, dtype, stats=st)
    torch.cuda.synchronize()
    NEV = 40
    buf = (C.c_ulonglong * (8 * NEV + 8))()
    fn = cabi.lib().ld_debug_ws_trace
    fn.restype, fn.argtypes = C.c_int, [C.POINTER(C.c_ulonglong)]
    assert fn(buf) == 0
    print(f"== {cin}->{cout}@{H}x{W} B{B} prologue={prologue}")
    print("  wave -> (simd, wave slot, cu):", [((buf[8 * NEV + i] >> 4) & 3, buf[8 * NEV + i] & 15, (buf[8 * NEV + i] >> 8) & 15) for i in range(8)])
    t0 = min(buf[w * NEV] for w in range(8))
    for w in range(8):
        ev = [buf[w * NEV + k] for k in range(NEV) if buf[w * NEV + k]]
        print(f"  wave {w} ({'cons' if w < 4 else 'prod'}) t(cyc since start):", " ".join(str(e - t0) for e in ev))

if __name__ == "__main__":
    run(8, 256, 256, 32, 32)
    run(8, 256, 256, 32, 32, prologue=True)
