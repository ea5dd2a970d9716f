"""Micro-benchmark of ld_conv3x3 on one shape (GPU box).  LD_CONV_DEBUG ablation bits apply."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import hip_helpers as hh
from localdiffusion_hallucination_amd import _cabi as cabi
import ctypes as C

def bench(B, cin, cout, H, W, dtype="bf16", stats=False, reps=50, prologue=False):
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
        out = hh.conv3x3([src], w, b, B, H, W, cout, dtype, stats=st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = hh.conv3x3([src], w, b, B, H, W, cout, dtype, stats=st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    es = 2 if dtype == "bf16" else 4
    byts = (B * H * W * (cin + cout)) * es
    print(f"dbg={os.environ.get('LD_CONV_DEBUG','0'):>2s} {cin}->{cout}@{H}x{W} B{B} {dtype} stats={stats} pro={prologue}: {us:8.1f} us  {byts/us/1e3:7.1f} GB/s  {2*9*cin*cout*B*H*W/us/1e6:7.1f} TF/s")

if __name__ == "__main__":
    shapes = [(8, 32, 32, 256, 256), (8, 64, 32, 256, 256), (8, 32, 32, 128, 128), (8, 64, 64, 128, 128),
              (8, 64, 64, 64, 64), (8, 128, 128, 64, 64), (8, 128, 128, 32, 32), (8, 256, 256, 32, 32), (8, 512, 256, 32, 32)]
    print("KSPLIT", os.environ.get("LD_CONV_KSPLIT", "-"), "MT", os.environ.get("LD_CONV_MT", "-"), "NW", os.environ.get("LD_CONV_NW", "-"), "NO_C32", os.environ.get("LD_CONV_NO_C32", "-"), "DB", os.environ.get("LD_CONV_DB", "-"))
    sel = shapes[:3] if not os.environ.get("LD_BENCH_SMALL") else shapes[3:]
    if os.environ.get("LD_BENCH_SHAPES"):          # "B,cin,cout,H,W;B,cin,cout,H,W;..."
        sel = [tuple(int(v) for v in t.split(",")) for t in os.environ["LD_BENCH_SHAPES"].split(";")]
    if os.environ.get("LD_BENCH_SEL"):
        sel = [shapes[int(i)] for i in os.environ["LD_BENCH_SEL"].split(",")]
    for shape in sel:
        bench(*shape)
        bench(*shape, stats=True)
        if os.environ.get("LD_BENCH_PRO"):
            bench(*shape, stats=True, prologue=True)
