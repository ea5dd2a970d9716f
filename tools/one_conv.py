"""GPU box: one conv3x3 shape launched many times (for rocprofv3 --pmc passes, tools/pmc_conv_stalls.sh).
usage: python3 tools/one_conv.py B cin cout H [iters] [prologue]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import hip_helpers as hh

B, cin, cout, H = (int(v) for v in sys.argv[1:5])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 200
prologue = len(sys.argv) > 6 and sys.argv[6] == "1"
dtype = "bf16"
x = torch.randn(B, H, H, cin, device="cuda").to(hh.TDT[dtype])
w = hh.pack(torch.randn(cout, cin, 3, 3) * 0.05, dtype, 3)
b = torch.zeros(cout, device="cuda")
st = hh.stats_buffer(B, 8)
if prologue:
    gn = (hh.stats_striped(x.float().permute(0, 3, 1, 2), 8), torch.ones(cin, device="cuda"), torch.zeros(cin, device="cuda"), 8)
    src = hh.make_src(x, cin, gn=gn, act=1)
else:
    src = hh.make_src(x, cin)
for _ in range(iters):
    hh.conv3x3([src], w, b, B, H, H, cout, dtype, stats=st)
torch.cuda.synchronize()
