#!/bin/bash
# GPU box: MFMA utilisation per kernel of the bench step (one batch on one stream).
#   rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES
# utilisation = MFMA busy cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)   (GRBM_GUI_ACTIVE is summed over the 8 XCDs)
# usage: bash tools/pmc_mfma.sh <tag>   -> gpurun_out/<tag>_pmc_mfma.txt
TAG=${1:-rXX}
REGIME=${2:-solo}      # solo: LD_SUB_BATCHES=1 (one batch, one stream) -> <tag>_pmc_mfma.txt;  default: the timed two-sub-batch regime -> <tag>_pmc_mfma_default.txt
if [ "$REGIME" = default ]; then SB=2; ST=; SUF=_default; else SB=1; ST=_s1; SUF=; fi
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_m
LD_SUB_BATCHES=$SB rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d /tmp/prof_m -o r -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-other-dtype > $OUT/${TAG}_pm.log 2>&1 < /dev/null
python3 - <<PY > $OUT/${TAG}_pmc_mfma${SUF}.txt
import csv, glob, collections, re, os
def short(k):
    k = re.sub(r"\(anonymous namespace\)::", "", k); return re.sub(r"_ZN12_GLOBAL__N_1\d+", "", k)[:72]
# un-profiled durations of the same kernels: the kernel-trace summary of the same command (profile_round.sh)
real = {}
st = "$OUT/${TAG}${ST}_kernel_stats.csv" if os.path.exists("$OUT/${TAG}${ST}_kernel_stats.csv") else "$R/profiles/${TAG}${ST}_kernel_stats.csv"
if os.path.exists(st):
    for r in csv.DictReader(open(st)): real[short(r["Name"])] = float(r["AverageNs"])
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("/tmp/prof_m/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE": cnt[k] += 1
print("MFMA utilisation per kernel, bench step, regime: $REGIME (LD_SUB_BATCHES=$SB: 1 = one batch of 8 on one stream, 2 = the timed regime, two concurrent sub-batches of 4), rocprofv3 --pmc")
print("MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs per launch; 'in window' divides by GRBM_GUI_ACTIVE / 8 (the counter window,")
print("which includes the profiler's per-dispatch overhead); 'vs peak' divides by the kernel's un-profiled duration")
print("(kernel-trace summary) x 2.4 GHz, i.e. the fraction of the 2.5 PFLOP/s bf16 peak; VALU = share of wave time issuing VALU")
print(f"{'kernel':74s} {'launches':>8s} {'us (trace)':>10s} {'MFMA in window':>15s} {'MFMA vs peak':>13s} {'VALU':>7s}")
rows = []
for k, d in agg.items():
    act = d.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    if act <= 0 or cnt[k] == 0: continue
    util = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * act)
    valu = d.get("SQ_ACTIVE_INST_VALU", 0.0) / max(d.get("SQ_WAVE_CYCLES", 1.0), 1.0)
    rows.append((act, k, cnt[k], util, valu))
for act, k, n, util, valu in sorted(rows, reverse=True)[:30]:
    if k.startswith("void at::") or "rocclr" in k: continue
    busy = agg[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0 / n
    ns = real.get(k)
    peak = f"{100 * busy / (ns * 2.4):12.1f}%" if ns else f"{'-':>13s}"
    print(f"{k:74s} {n:8d} {(ns or 0) / 1e3:10.1f} {100 * util:14.1f}% {peak} {100 * valu:6.1f}%")
PY
cat $OUT/${TAG}_pmc_mfma${SUF}.txt
