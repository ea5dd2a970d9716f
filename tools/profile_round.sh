#!/bin/bash
# GPU box: the evidence set of one round.  usage: bash tools/profile_round.sh r01_d
#   1. default bench line                      -> gpurun_out/<tag>_bench.json
#   2. rocprofv3 --kernel-trace --stats of the same command (fewer steps) -> <tag>_kernel_stats.csv
#   3. separate --pmc FETCH_SIZE and --pmc WRITE_SIZE passes -> <tag>_pmc_traffic.json / _pmc_summary.txt
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $OUT/${TAG}_bench.log 2>&1 < /dev/null
tail -1 $OUT/${TAG}_bench.log > $OUT/${TAG}_bench.json
rm -rf /tmp/prof_ks /tmp/prof_f /tmp/prof_w
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ks -o r -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $OUT/${TAG}_ks.log 2>&1 < /dev/null
cp $(find /tmp/prof_ks -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_kernel_stats.csv
# the same two for ONE batch on one stream (LD_SUB_BATCHES=1): rocprofv3's interception slows the graph launches of the
# default two-sub-batch regime (its kernels then overlap less than un-profiled), the single-stream eager run is not affected
LD_SUB_BATCHES=1 python3 $R/bench.py --no-cpu-baseline > $OUT/${TAG}_s1_bench.log 2>&1 < /dev/null
tail -1 $OUT/${TAG}_s1_bench.log > $OUT/${TAG}_s1_bench.json
rm -rf /tmp/prof_ks1
LD_SUB_BATCHES=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ks1 -o r -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $OUT/${TAG}_s1_ks.log 2>&1 < /dev/null
cp $(find /tmp/prof_ks1 -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_s1_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_f -o r -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/${TAG}_pf.log 2>&1 < /dev/null
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_w -o r -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/${TAG}_pw.log 2>&1 < /dev/null
python3 $R/tools/pmc_summarize.py /tmp/prof_f /tmp/prof_w $OUT/${TAG}_pmc_traffic.json > $OUT/${TAG}_pmc_summary.txt 2>&1
cut -c1-200 $OUT/${TAG}_bench.json
head -14 $OUT/${TAG}_kernel_stats.csv | cut -c1-150
tail -15 $OUT/${TAG}_pmc_summary.txt
