"""Per-configuration median kernel duration (us) from the rocprofv3 kernel traces written by ablate_conv.sh."""
import csv, glob, os, re, statistics as st, sys
out = sys.argv[1]
table, labels, bits = {}, [], []
for d in sorted(glob.glob(os.path.join(out, "d*")), key=lambda p: int(os.path.basename(p)[1:])):
    dbg = int(os.path.basename(d)[1:])
    f = glob.glob(os.path.join(d, "*kernel_trace.csv"))
    if not f:
        continue
    rows = [r for r in csv.DictReader(open(f[0])) if "conv3x3" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    cfgs = [l for l in open(os.path.join(out, f"log_{dbg}.txt")) if l.startswith("dbg=")]
    per = 55                                   # 5 warm-up + 50 timed launches per configuration
    if len(rows) != per * len(cfgs):
        print(f"dbg={dbg}: {len(rows)} launches for {len(cfgs)} configurations, skipped"); continue
    bits.append(dbg)
    for i, l in enumerate(cfgs):
        lab = re.sub(r"^dbg=\s*\d+\s*", "", l.split(":")[0])
        if lab not in labels: labels.append(lab)
        dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[i * per + 5:(i + 1) * per]]
        kn = re.sub(r".*(conv3x3\w*kernel)I(\w+?)E.*", r"\1<\2>", rows[i * per]["Kernel_Name"])[:34]
        table[(dbg, lab)] = (st.median(dur) / 1e3, kn)
print("config".ljust(50) + "kernel".ljust(36) + "".join(f"dbg{b:>3d} " for b in bits))
for lab in labels:
    print(lab.ljust(50) + table[(bits[0], lab)][1].ljust(36) + "".join(f"{table[(b, lab)][0]:6.1f} " for b in bits if (b, lab) in table))
