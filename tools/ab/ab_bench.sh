#!/bin/bash
# GPU box: same-box A/B of the working tree's library against tools/ab/libold.so (tools/ab/build_old.sh <rev>), alternating runs
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for which in new old; do
    if [ $which = old ]; then export LD_LIB_OVERRIDE=$GRAFT_REPO_ROOT/tools/ab/libold.so; else unset LD_LIB_OVERRIDE; fi
    python bench.py --no-cpu-baseline --no-other-dtype "$@" 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$which', round(d['ms_per_step'],4), 'solo-sum', r['step_ms_sum_of_kernels'], 'in-situ-sum', r.get('in_situ',{}).get('step_ms_sum_of_kernels'))"
  done
done
