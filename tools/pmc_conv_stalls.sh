#!/bin/bash
# GPU box: what the waves of one conv3x3 launch wait for (SQ counters, DESIGN finding 43).
# usage: bash tools/pmc_conv_stalls.sh "8 256 256 32" [tag]
SHAPE=${1:-"8 256 256 32"}
TAG=${2:-conv}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export LD_CONV_NO_C32=1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT" \
           "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_SMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/pc_$i
  rocprofv3 --pmc $set --output-format csv -d /tmp/pc_$i -o r -- python3 $R/tools/one_conv.py $SHAPE 40 > /tmp/pc_$i.log 2>&1 < /dev/null || tail -3 /tmp/pc_$i.log
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob("/tmp/pc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv3x3_kernel" not in r["Kernel_Name"]: continue
        agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
v = {k: agg[k] / n[k] for k in agg}
print("shape (B cin cout H): $SHAPE   per launch, summed over the chip")
for k in sorted(v): print(f"  {k:32s} {v[k]:16.0f}")
wc = v.get("SQ_WAVE_CYCLES", 1.0)
print(f"waves {v.get('SQ_WAVES', 0):.0f}; wave cycles per wave {wc / max(v.get('SQ_WAVES', 1), 1):.0f} (x4: the counter ticks every 4 clocks?)")
for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_VALU",
          "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC", "SQ_ACTIVE_INST_FLAT"):
    if k in v: print(f"  {k:24s} {100 * v[k] / wc:6.1f} % of wave cycles")
if "SQ_INSTS_VMEM_RD" in v: print(f"  VMEM read latency  ~ {v['SQ_INST_LEVEL_VMEM'] / max(v['SQ_INSTS_VMEM_RD'], 1):.0f} cycles (INST_LEVEL_VMEM / INSTS_VMEM_RD; includes writes in the level)")
if "SQ_INSTS_LDS" in v: print(f"  LDS latency        ~ {v['SQ_INST_LEVEL_LDS'] / max(v['SQ_INSTS_LDS'], 1):.0f} cycles; bank conflict cycles / LDS active = {v.get('SQ_LDS_BANK_CONFLICT', 0) / max(v.get('SQ_LDS_IDX_ACTIVE', 1), 1):.3f}")
PY
