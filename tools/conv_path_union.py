"""Chip-level view of the ResBlock conv path in the timed regime (two concurrent sub-batches), from a rocprofv3 kernel trace
of the default bench command (tools/profile_round.sh).

`roofline.in_situ.resblock_conv_path.hbm_frac` prices ONE sub-batch's launches while the other sub-batch shares the chip.
The launches of the path are the twelve conv3x3<*,2,4> dispatches of a sub-batch step (8 x 32->32 and 4 x 64->32 at 256^2,
4 patches each: 444.9 MB of algorithmic traffic per sub-batch step).  This script takes the UNION of their execution
intervals over both queues: path bytes of both sub-batches / union time = what the chip moves for the path while any of its
kernels is running (other kernels of the other stream may run beside them, so this is a lower bound of the traffic in
those intervals).  usage: python tools/conv_path_union.py <kernel_trace.csv> [bytes per sub-batch step]"""
import csv, sys
path = sys.argv[1]
bytes_per_sub_step = float(sys.argv[2]) if len(sys.argv) > 2 else 444892672.0
rows = list(csv.DictReader(open(path)))
name_key = next(k for k in rows[0] if k.lower() in ("kernel_name", "name"))
s_key = next(k for k in rows[0] if k.lower().startswith("start"))
e_key = next(k for k in rows[0] if k.lower().startswith("end"))
# (round 5: the single-chunk launches of the path run on conv3x3_s32_kernel, the two-chunk ones still on conv3x3<2,4>)
sel = [(int(r[s_key]), int(r[e_key])) for r in rows
       if ("conv3x3_kernel" in r[name_key] and "Li2ELi4E" in r[name_key]) or "conv3x3_s32_kernel" in r[name_key]]
sel.sort()
if not sel:
    print("no conv-path dispatches in", path); sys.exit(0)
# drop the encoder / warm-up part: keep the last 80 % of the dispatches
sel = sel[len(sel) // 5:]
union, cur_s, cur_e, busy_sum = 0, sel[0][0], sel[0][1], 0
for s, e in sel:
    busy_sum += e - s
    if s > cur_e:
        union += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
n = len(sel)
total_bytes = n / 12.0 * bytes_per_sub_step
print(f"{n} conv-path dispatches (conv3x3_s32 + conv3x3<2,4>; {n / 24:.1f} steps of two sub-batches), sum of durations {busy_sum / 1e3:.1f} us, union {union / 1e3:.1f} us "
      f"(overlap factor {busy_sum / union:.2f})")
print(f"per launch: {busy_sum / n / 1e3:.2f} us = {bytes_per_sub_step / 12 / (busy_sum / n):.3f} GB/ms = "
      f"{bytes_per_sub_step / 12 / (busy_sum / n) / 8000 * 100:.1f} % of 8 TB/s")
print(f"chip level: {total_bytes / 1e6:.0f} MB in {union / 1e3:.0f} us = {total_bytes / union:.3f} GB/ms = {total_bytes / union / 8000 * 100:.1f} % of 8 TB/s "
      f"(algorithmic bytes of the path only; rocprofv3 slows the graph replays, so un-profiled overlap is higher)")
