"""Summarise rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE collected in SEPARATE runs) into per-kernel
HBM traffic per launch, with the gfx950 corrections of MI355X_MICROARCH.md (HBM section):
FETCH_SIZE counts 64 B per 128-B request of a wide coalesced stream -> doubled; WRITE_SIZE is exact;
both are reported in KiB by rocprofv3.  Usage: pmc_summarize.py <fetch_dir> <write_dir> <out.json>"""
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
    bf = "DF16b" in name or "bf16" in name or "_Accum" in name     # rocprofv3's demangler prints __bf16 as "bool _Accum"
    if "conv3x3_c32_kernel" in name:                           # persistent variant of the <.,2,4> family
        return f"conv3x3<{'bf16' if bf else 'f32'},2,4>"
    m = re.search(r"conv3x3_kernelI(?:f|DF16b)Li(\d)ELi(\d)E", name) or re.search(r"conv3x3_kernel<[^,]+, (\d), (\d)", name)
    if m:
        return f"conv3x3<{'bf16' if bf else 'f32'},{m.group(1)},{m.group(2)}>"
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
json.dump(out, open(sys.argv[3], "w"), indent=1)
for f, d in sorted(out.items()):
    print(f"{f:24s} launches {d['launches']:6d}  read {d['read_bytes_per_launch']/1e6:9.2f} MB  write {d['write_bytes_per_launch']/1e6:9.2f} MB")
