#!/bin/bash
# GPU box: everything that is committed under profiles/<tag>_* for one tree, from ONE box:
# the -m gpu suite + smoke(), tools/profile_round.sh (bench lines, kernel stats default + solo, PMC traffic),
# tools/pmc_mfma.sh, and the extra bench lines (driver's --steps 20, 64 patches per GPU, cfg5, fp16 with two-term weights).
# usage: bash tools/evidence_round.sh r03_d
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
cd $R
python -m pytest tests -x -q -m gpu > $OUT/${TAG}_tests.log 2>&1; tail -3 $OUT/${TAG}_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/${TAG}_smoke.log 2>&1; tail -1 $OUT/${TAG}_smoke.log
(grep -E "passed|failed" $OUT/${TAG}_tests.log; tail -1 $OUT/${TAG}_smoke.log) > $OUT/${TAG}_gpu_tests.txt
bash tools/profile_round.sh $TAG > $OUT/${TAG}_profile_round.log 2>&1
bash tools/pmc_mfma.sh $TAG >> $OUT/${TAG}_profile_round.log 2>&1
bash tools/pmc_mfma.sh $TAG default >> $OUT/${TAG}_profile_round.log 2>&1
cd /tmp
python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-dtype 2> /dev/null | tail -1 > $OUT/${TAG}_bench_steps20.json
# the complete default line exactly as the driver runs it (every leg: roofline, other dtype, G16 distance, CPU baseline)
python3 $R/bench.py --steps 20 --warmup 5 2> /dev/null | tail -1 > $OUT/${TAG}_bench_driver.json
python3 $R/bench.py --patches 64 --steps 60 --no-cpu-baseline --no-other-dtype 2> /dev/null | tail -1 > $OUT/${TAG}_p64_bench.json
python3 $R/bench.py --dtype fp16 --weight-split-levels 2 --no-cpu-baseline --no-other-dtype --no-roofline 2> /dev/null | tail -1 > $OUT/${TAG}_fp16x2_bench.json
for f in bench s1_bench fp16_bench bench_steps20 bench_driver p64_bench cfg5_bench fp16x2_bench; do python3 -c "
import json,sys; d=json.load(open('$OUT/${TAG}_$f.json')); print('%-16s' % '$f', d['value'], d['unit'], d['ms_per_step'], 'ms/step', d.get('roofline',{}).get('frac'))"; done
