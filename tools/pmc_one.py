"""Run one conv3x3 shape a few times (for rocprofv3 --pmc)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, hip_helpers as hh
B, cin, cout, H, W = [int(v) for v in sys.argv[1:6]]
x = torch.randn(B, H, W, cin, device="cuda").to(torch.bfloat16)
w = hh.pack(torch.randn(cout, cin, 3, 3) * 0.05, "bf16", 3)
b = torch.zeros(cout, device="cuda")
src = hh.make_src(x, cin)
for _ in range(6):
    out = hh.conv3x3([src], w, b, B, H, W, cout, "bf16")
torch.cuda.synchronize()
