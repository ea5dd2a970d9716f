"""Micro-benchmark of ld_attention (GPU box): (B, n) cases, one- and two-key-group routing (attn_split_max_wgs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import hip_helpers as hh
from localdiffusion_hallucination_amd import _cabi as cabi

lib = cabi.lib()
cabi.check(lib.ld_tuning_set(b"attn_split_min_n", 256), "set")
for dtype in ("bf16",):
    for B, n in ((4, 1024), (8, 1024), (1, 4096), (2, 4096)):
        qkv = (torch.randn(B, n, 384, device="cuda") * 0.5).to(hh.TDT[dtype])
        out = torch.empty(B, n, 128, dtype=hh.TDT[dtype], device="cuda")
        for wgs in (0, 1 << 30):
            cabi.check(lib.ld_tuning_set(b"attn_split_max_wgs", wgs), "set")
            for _ in range(5):
                lib.ld_attention(qkv.data_ptr(), out.data_ptr(), B, n, 4, 32, cabi.dtype_code(dtype), hh.st())
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200):
                lib.ld_attention(qkv.data_ptr(), out.data_ptr(), B, n, 4, 32, cabi.dtype_code(dtype), hh.st())
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 200
            fl = 4.0 * B * 4 * n * n * 32
            print(f"{dtype} B={B} n={n} {'two groups' if wgs else 'one group '}: {us:7.2f} us  {fl / us / 1e6:7.1f} TFLOP/s")
