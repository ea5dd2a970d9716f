"""GPU box: cycle stamps of one workgroup of the persistent C=32 conv (LD_CONV_DEBUG bit 32)."""
import os, sys, ctypes as C
os.environ["LD_CONV_DEBUG"] = str(32 | int(os.environ.get("LD_TRACE_EXTRA", "0")))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import hip_helpers as hh
from localdiffusion_hallucination_amd import _cabi as cabi

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
    buf = (C.c_ulonglong * 16)()
    fn = cabi.lib().ld_debug_c32_trace
    fn.restype, fn.argtypes = C.c_int, [C.POINTER(C.c_ulonglong)]
    assert fn(buf) == 0
    t = [buf[k] for k in range(16)]
    names = ["start", "setup", "ringfill", "i3:top", "i3:dma", "i3:phase1", "-", "-", "i3:phase2", "i3:barrier", "-", "-", "-", "loop_end", "end"]
    print(f"== {cin}->{cout}@{H}x{W} B{B} prologue={prologue} stats={stats} dbg={os.environ['LD_CONV_DEBUG']}")
    print("  " + " ".join(f"{n}:{t[k] - t[0]}" for k, n in enumerate(names) if t[k]))

if __name__ == "__main__":
    for B in [int(v) for v in os.environ.get("LD_TRACE_B", "8").split(",")]:
        run(B, 32, 32, 256, 256, stats=False)
        run(B, 32, 32, 256, 256, stats=True)
        run(B, 32, 32, 256, 256, prologue=True)
