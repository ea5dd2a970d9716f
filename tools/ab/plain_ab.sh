#!/bin/bash
# GPU box: the working tree's library (write-through output stores) against tools/ab/libplain.so (tools/ab/build_plain.sh),
# alternating runs of the default bench at 400 and 20 steps + cfg5.  usage: bash tools/ab/plain_ab.sh [rounds]
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/wt
export PYTHONUNBUFFERED=1
N=${1:-3}
for i in $(seq 1 $N); do
  for which in wt plain; do
    if [ $which = plain ]; then export LD_LIB_OVERRIDE=$GRAFT_REPO_ROOT/tools/ab/libplain.so; else unset LD_LIB_OVERRIDE; fi
    python bench.py --no-cpu-baseline --no-other-dtype --no-roofline --steps 400 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$which steps=400', round(d['ms_per_step'],4))" >> gpurun_out/wt/ab.txt
    python bench.py --no-cpu-baseline --no-other-dtype --no-roofline --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$which steps=20', round(d['ms_per_step'],4))" >> gpurun_out/wt/ab.txt
    python bench.py --workload cfg5 --no-roofline --steps 500 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$which cfg5', round(d['value'],3), round(d['ms_per_step'],4))" >> gpurun_out/wt/ab.txt
  done
done
unset LD_LIB_OVERRIDE
timeout 1200 python -m pytest tests/test_hip_ops.py tests/test_hip_unet.py tests/test_hip_bench_shape.py -q -x > gpurun_out/wt/tests.txt 2>&1
tail -2 gpurun_out/wt/tests.txt
echo done
