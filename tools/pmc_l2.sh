#!/bin/bash
# GPU box: L2 -> L1 read traffic and L2 hit rate per kernel of the bench step (one batch on one stream), DESIGN finding 41.
#   pass A: TCP_TCC_READ_REQ_sum (vL1D -> L2 read requests, 64 B each), TCP_TCC_READ_REQ_LATENCY_sum
#   pass B: TCC_HIT_sum TCC_MISS_sum
# usage: bash tools/pmc_l2.sh <tag>   -> gpurun_out/<tag>_pmc_l2.txt   (durations: profiles/<tag>_s1_kernel_stats.csv)
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_la /tmp/prof_lb
LD_SUB_BATCHES=1 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d /tmp/prof_la -o r -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-other-dtype > $OUT/${TAG}_la.log 2>&1 < /dev/null
LD_SUB_BATCHES=1 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d /tmp/prof_lb -o r -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-other-dtype > $OUT/${TAG}_lb.log 2>&1 < /dev/null
python3 - <<PY > $OUT/${TAG}_pmc_l2.txt
import csv, glob, collections, re, os
def short(k):
    k = re.sub(r"\(anonymous namespace\)::", "", k); return re.sub(r"_ZN12_GLOBAL__N_1\d+", "", k)[:72]
real = {}
for st in ("$OUT/${TAG}_s1_kernel_stats.csv", "$R/profiles/${TAG}_s1_kernel_stats.csv"):
    if os.path.exists(st):
        for r in csv.DictReader(open(st)): real[short(r["Name"])] = float(r["AverageNs"])
        break
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for d in ("/tmp/prof_la", "/tmp/prof_lb"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
print("L2 -> vL1D read traffic per kernel, bench step as one batch of 8 on one stream (LD_SUB_BATCHES=1), rocprofv3 --pmc")
print("MB = TCP_TCC_READ_REQ_sum x 64 B per launch; TB/s = MB / un-profiled duration (kernel-trace summary);")
print("latency = TCP_TCC_READ_REQ_LATENCY_sum / TCP_TCC_READ_REQ_sum (cycles, under the profiler); hit = TCC_HIT / (TCC_HIT + TCC_MISS)")
print(f"{'kernel':74s} {'launches':>8s} {'us':>7s} {'L2->L1 MB':>10s} {'TB/s':>7s} {'latency':>8s} {'L2 hit':>7s}")
rows = []
for k, d in agg.items():
    n = cnt[k].get("TCP_TCC_READ_REQ_sum", 0)
    if not n or k.startswith("void at::") or "rocclr" in k: continue
    mb = d["TCP_TCC_READ_REQ_sum"] * 64 / n / 1e6
    lat = d["TCP_TCC_READ_REQ_LATENCY_sum"] / max(d["TCP_TCC_READ_REQ_sum"], 1.0)
    hit = d.get("TCC_HIT_sum", 0.0) / max(d.get("TCC_HIT_sum", 0.0) + d.get("TCC_MISS_sum", 0.0), 1.0)
    us = real.get(k, 0.0) / 1e3
    rows.append((mb * n, k, n, us, mb, lat, hit))
for _, k, n, us, mb, lat, hit in sorted(rows, reverse=True)[:32]:
    print(f"{k:74s} {n:8d} {us:7.1f} {mb:10.1f} {mb / us if us else 0.0:7.2f} {lat:8.0f} {100 * hit:6.1f}%")
PY
cat $OUT/${TAG}_pmc_l2.txt
