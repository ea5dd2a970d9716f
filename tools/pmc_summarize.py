"""Summarise rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE collected in SEPARATE runs) into per-kernel
HBM traffic per launch, with the gfx950 corrections of MI355X_MICROARCH.md (HBM section):
FETCH_SIZE counts 64 B per 128-B request of a wide coalesced stream (16 B per lane) -> doubled; WRITE_SIZE is exact;
both are reported in KiB by rocprofv3.  Every global load of these kernels is a 16-byte-per-lane load of 64-byte
(pixel) or 1-KiB (weight block) contiguous runs; the doubling is CALIBRATED inside the same passes by the kernels
whose traffic is known exactly (gn_apply: reads each operand once and writes the result once; linattn_kvctx: reads x
once) -- the summary prints their counter / algorithmic ratio, which must come out at 1.0 for the correction to hold.
Usage: pmc_summarize.py <fetch_dir> <write_dir> <out.json> [bench.json with roofline.families]"""
import csv, glob, json, os, sys
from collections import defaultdict


def load(d, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            k = r["Kernel_Name"]
            acc[k][0] += float(r["Counter_Value"])
            acc[k][1] += 1
    return acc


def family(name):
    import re
    # storage type from the (mangled or demangled) name; rocprofv3's demangler prints __bf16 as "bool _Accum"
    dt = "bf16" if ("DF16b" in name or "bf16" in name or "_Accum" in name) else ("f16" if ("DF16_" in name or "_Float16" in name) else "f32")
    if "conv3x3_c32_kernel" in name:                           # persistent LDS-DMA variant (bench.py names it the same)
        return f"conv3x3_c32<{dt}>"
    if "conv3x3_s32_kernel" in name:                           # the lean Cout = 32 kernel (conv3x3_s32.hip)
        return f"conv3x3_s32<{dt}>"
    m = re.search(r"conv3x3_kernelI(?:f|DF16b|DF16_)Li(\d)ELi(\d)E", name) or re.search(r"conv3x3_kernel<[^,]+, (\d), (\d)", name)
    if m:
        return f"conv3x3<{dt},{m.group(1)},{m.group(2)}>"
    for key, fam in [("conv1x1", "conv1x1"), ("gn_apply", "gn_apply"), ("kvctx", "linattn_kvctx"), ("linout", "linattn_out"),
                     ("attention", "attention"), ("conv_image", "conv_image7x7"), ("conv_stem", "conv_image7x7"),
                     ("ctxfold", "linattn_ctxfold")]:
        if key in name:
            return fam
    return None


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
fams = defaultdict(lambda: dict(fetch_kib=0.0, write_kib=0.0, launches=0))
for k, (v, n) in fetch.items():
    f = family(k)
    if f:
        fams[f]["fetch_kib"] += v
        fams[f]["launches"] += n
for k, (v, n) in write.items():
    f = family(k)
    if f:
        fams[f]["write_kib"] += v
for f, d in fams.items():
    n = max(1, d["launches"])
    rd = 2.0 * d["fetch_kib"] * 1024 / n          # gfx950: FETCH_SIZE reports half of a wide coalesced stream
    wr = d["write_kib"] * 1024 / n
    out[f] = {"hbm_bytes_per_launch": rd + wr, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "launches": n,
              "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE x2 (gfx950 correction)"}
algo = {}
if len(sys.argv) > 4:
    try:
        algo = json.loads(open(sys.argv[4]).read().strip().splitlines()[-1])["roofline"]["families"]
    except Exception as e:                                       # the summary is still useful without the comparison
        print(f"(no algorithmic bytes: {e})")
for f, d in out.items():
    if f in algo:
        d["algorithmic_bytes_per_launch"] = algo[f]["bytes_per_launch"]
        d["traffic_over_algorithmic"] = round(d["hbm_bytes_per_launch"] / max(1, algo[f]["bytes_per_launch"]), 3)
json.dump(out, open(sys.argv[3], "w"), indent=1)
for f, d in sorted(out.items()):
    extra = f"  algorithmic {d['algorithmic_bytes_per_launch']/1e6:9.2f} MB  ratio {d['traffic_over_algorithmic']:.2f}" if "traffic_over_algorithmic" in d else ""
    print(f"{f:24s} launches {d['launches']:6d}  read {d['read_bytes_per_launch']/1e6:9.2f} MB  write {d['write_bytes_per_launch']/1e6:9.2f} MB{extra}")
