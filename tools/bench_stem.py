"""Micro-benchmark of ld_conv_stem (init_conv 7x7, 16-bit storage) alone on the chip (GPU box).
usage: python tools/bench_stem.py ["B,H,W;B,H,W;..."]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import hip_helpers as hh
from localdiffusion_hallucination_amd import _cabi as cabi


def bench(B, H, W, cin=3, dtype="bf16", reps=100):
    lib = cabi.lib()
    x = torch.randn(B, cin, H, W, device="cuda")
    w = (torch.randn(32, cin, 7, 7, device="cuda") * 0.1).contiguous()
    b = torch.zeros(32, device="cuda")
    wp = torch.empty(int(lib.ld_stem_packed_bytes()), dtype=torch.uint8, device="cuda")
    cabi.check(lib.ld_pack_stem_weight(w.data_ptr(), wp.data_ptr(), cin, hh.st()), "pack_stem")
    out = torch.empty(B, H, W, 32, dtype=hh.TDT[dtype], device="cuda")
    def run():
        cabi.check(lib.ld_conv_stem(x.data_ptr(), wp.data_ptr(), b.data_ptr(), out.data_ptr(), B, cin, H, W, cabi.dtype_code(dtype), hh.st()), "conv_stem")
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    byts = B * H * W * (cin * 4 + 32 * 2)
    print(f"stem {cin}x{H}x{W} B{B} {dtype}: {us:8.1f} us  {byts / us / 1e3:7.1f} GB/s  {2 * 49 * cin * 32 * B * H * W / us / 1e6:7.1f} TF/s (useful)")


if __name__ == "__main__":
    shapes = sys.argv[1] if len(sys.argv) > 1 else "4,256,256;8,256,256;32,256,256;64,256,256"
    for t in shapes.split(";"):
        B, H, W = (int(v) for v in t.split(","))
        bench(B, H, W)
    bench(1, 512, 512, cin=1, dtype="fp16")
