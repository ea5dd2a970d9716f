"""GPU box: do the two sub-batches' ResBlock-conv-path launches really run at the same time INSIDE the replayed step graphs,
and what does the chip move for the path while they do?  (`roofline.in_situ.resblock_conv_path.hbm_frac` prices ONE
sub-batch's launches while the other shares the chip.)

Needs tools/ab/libdbg.so (tools/ab/build_dbg.sh) and LD_CONV_DEBUG=128: every 3x3 convolution launch of the generic kernel
owns a slot and leaves its start (workgroup 0) and its latest end (any workgroup) on the 100 MHz real-time clock there
(csrc/conv3x3_body.hip.h).  The slots are baked into the captured graphs, one clock serves both streams, and after N
replayed steps they hold the LAST step's positions: per-launch durations, the union of the intervals over both
sub-batches, path bytes / union time.  The stamps cost one atomic per workgroup.
usage: LD_LIB_OVERRIDE=$PWD/tools/ab/libdbg.so LD_CONV_DEBUG=128 python tools/exp_conv_path_overlap.py [patches per GPU]"""
import ctypes as C
import os, sys
os.environ.setdefault("LD_CONV_DEBUG", "128")
os.environ.setdefault("LD_CONV_NO_C32", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import localdiffusion_hallucination_amd as ldh
from localdiffusion_hallucination_amd import weights, _cabi as cabi

ldh.configure_runtime()
dev = torch.device("cuda:0")
H, P, T = 256, int(sys.argv[1]) if len(sys.argv) > 1 else 8, 1000
net = ldh.Unet(dim=32, init_dim=32, channels=3, out_dim=3, mode="mvtec", compute_dtype="bf16")
net.load_state_dict({k: torch.from_numpy(v) for k, v in weights.procedural_state_dict(net.cfg, 0).items()})
net = net.to(dev)
config = dict(branch_out=False, start_intermediate=False, start_timestep=2, data="mvtec", mask_x=False,
              ood_AD=False, ood_confidence=False, classifier=False, use_gt=False)
gd = ldh.GaussianDiffusion(config, net, image_size=H, timesteps=T, objective="pred_x0", beta_schedule="sigmoid").to(dev)
gd.noise_source = "device"
jp = net.plan(P, H, H, table_T=T)
jp.cond_in.uniform_(0.0, 2.0); jp.x_in.normal_()
z = torch.empty_like(jp.x_in)
lib = cabi.lib()
fn = lib.ld_debug_conv_spans
fn.restype, fn.argtypes = C.c_int, [C.POINTER(C.c_ulonglong), C.POINTER(C.c_int)]
spans, shapes = (C.c_ulonglong * 2048)(), (C.c_int * 5120)()

gd.encode_cond(jp, 60)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
draw = gd.run_joint_steps(jp, T - 1, 20, 0.0, 2.0, z, 1)       # captures the step graphs (slots baked in), warms up
e0.record()
draw = gd.run_joint_steps(jp, T - 21, 40, 0.0, 2.0, z, draw)
e1.record(); torch.cuda.synchronize()
step_us = 1e3 * e0.elapsed_time(e1) / 40
print(f"{P} patches as {gd.sub_batches} sub-batches, 40 replayed steps: {step_us / 1e3:.4f} ms per step (with the stamps)")
assert fn(spans, shapes) == 0
rows = []
for s in range(1024):
    a, b = spans[2 * s], spans[2 * s + 1]
    B, Hh, Ww, cin, cout = (shapes[5 * s + k] for k in range(5))
    if a and b > a:
        rows.append((a * 0.01, b * 0.01, B, Hh, Ww, cin, cout))          # 100 MHz ticks -> us
last_end = max(r[1] for r in rows)
rows = [r for r in rows if r[1] > last_end - 1.7 * step_us]    # the last step of both sub-batches (and the tail of the one before)


def report(what, sel):
    iv = sorted((a, b) for a, b, *_ in sel)
    nbytes = sum(B * Hh * Ww * (cin + cout) * 2 for _, _, B, Hh, Ww, cin, cout in sel)
    d = sum(b - a for a, b in iv)
    u, cs, ce = 0.0, iv[0][0], iv[0][1]
    for a, b in iv:
        if a > ce:
            u += ce - cs; cs, ce = a, b
        else:
            ce = max(ce, b)
    u += ce - cs
    print(f"{what}: {len(iv)} launches, {nbytes / 1e6:.0f} MB algorithmic; sum of durations {d:.1f} us ({d / len(iv):.2f} per launch = "
          f"{nbytes / d / 1e3 / 8000 * 100:.1f} % of 8 TB/s per launch), union {u:.1f} us (overlap factor {d / u:.2f}) -> chip level "
          f"{nbytes / u / 1e3:.0f} GB/s = {nbytes / u / 1e3 / 8000 * 100:.1f} % of 8 TB/s")


path = [r for r in rows if r[3] == H and r[4] == H and r[6] == 32]
# keep the last 12 launches per sub-batch: the slots also hold older steps' launches of other shapes
path.sort(key=lambda r: r[1])
path = path[-24:]
report("ResBlock conv path (C = 32 3x3 convolutions at 256^2), last replayed step", path)
small = [r for r in rows if r[3] <= 64]
small.sort(key=lambda r: r[1])
report("3x3 convolutions on the <= 64^2 maps, last replayed step(s)", small[-46:])
t00 = min(r[0] for r in path)
print("  conv path [start-end us]: " + " ".join(f"[{a - t00:.0f}-{b - t00:.0f}]" for a, b, *_ in sorted(path)))
