"""GPU box: ld_conv_image 7x7 stem at the cfg3 shape (run under rocprofv3 --kernel-trace for true durations)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import hip_helpers as hh
from localdiffusion_hallucination_amd import _cabi as cabi
B, H, W = 8, 256, 256
x = torch.randn(B, 3, H, W, device="cuda")
w = (torch.randn(32, 3, 7, 7, device="cuda") * 0.1).contiguous()
b = torch.zeros(32, device="cuda")
out = torch.empty(B, H, W, 32, dtype=torch.bfloat16, device="cuda")
lib = cabi.lib()
wp = torch.empty(int(lib.ld_stem_packed_bytes()), dtype=torch.uint8, device="cuda")
cabi.check(lib.ld_pack_stem_weight(w.data_ptr(), wp.data_ptr(), 3, hh.st()))
valu = bool(os.environ.get("LD_STEM_VALU"))
def call():
    if valu:
        cabi.check(lib.ld_conv_image(x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), None, 0, B, 3, H, W, 7, cabi.LD_BF16, hh.st()))
    else:
        cabi.check(lib.ld_conv_stem(x.data_ptr(), wp.data_ptr(), b.data_ptr(), out.data_ptr(), B, 3, H, W, hh.st()))
for _ in range(5):
    call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    call()
e1.record(); torch.cuda.synchronize()
print(f"stem 7x7 8x3x256x256 -> bf16: {e0.elapsed_time(e1) * 1e3 / 50:.1f} us per launch")
