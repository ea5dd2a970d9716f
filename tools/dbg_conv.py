import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, torch.nn.functional as F
import hip_helpers as hh

def run(name, x, w, b, dtype):
    B, cin, H, W = x.shape; cout = w.shape[0]
    ref = F.conv2d(x, w, b, padding=1)
    out = hh.nchw(hh.conv3x3([hh.make_src(hh.nhwc(x, dtype), cin)], hh.pack(w, dtype, 3), b.to(hh.DEV), B, H, W, cout, dtype))
    d = (out - ref).abs()
    print(f"{name:28s} {dtype}: maxerr {float(d.max()):.3e} refmax {float(ref.abs().max()):.3e}", end="")
    if d.max() > 1e-3:
        idx = (d > 1e-3).nonzero()
        print(f"  bad {len(idx)}/{d.numel()} first {idx[:4].tolist()}")
        bad_c = sorted(set(idx[:, 1].tolist())); bad_y = sorted(set(idx[:, 2].tolist())); bad_x = sorted(set(idx[:, 3].tolist()))
        print(f"     bad channels {bad_c[:40]}\n     bad rows {bad_y[:40]}\n     bad cols {bad_x[:40]}")
        i = idx[0].tolist()
        print("     got", out[i[0], i[1], i[2], :8].tolist(), "\n     ref", ref[i[0], i[1], i[2], :8].tolist())
    else:
        print()

for dtype in ("fp32", "bf16"):
    B, cin, cout, H, W = 1, 32, 32, 16, 16
    x = hh.rand((B, cin, H, W), 1).to(hh.TDT[dtype]).float()
    b = hh.rand((cout,), 3)
    w0 = torch.zeros(cout, cin, 3, 3)
    run("zero weights", x, w0, b, dtype)
    w = w0.clone()
    for c in range(32): w[c, c, 1, 1] = 1.0
    run("center identity", x, w, torch.zeros(cout), dtype)
    for (ky, kx) in [(0, 1), (1, 0), (2, 2)]:
        w = w0.clone()
        for c in range(32): w[c, c, ky, kx] = 1.0
        run(f"tap ({ky},{kx}) identity", x, w, torch.zeros(cout), dtype)
    w = w0.clone()
    for c in range(32): w[c, (c + 5) % 32, 1, 1] = 1.0
    run("center perm +5", x, w, torch.zeros(cout), dtype)
    w = hh.rand((cout, cin, 3, 3), 2, -0.1, 0.1).to(hh.TDT[dtype]).float()
    run("random", x, w, b, dtype)
    x2 = hh.rand((1, 64, 8, 8), 5).to(hh.TDT[dtype]).float()
    w2 = hh.rand((64, 64, 3, 3), 6, -0.1, 0.1).to(hh.TDT[dtype]).float()
    run("random 64->64 8x8", x2, w2, hh.rand((64,), 7), dtype)
