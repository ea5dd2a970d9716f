#!/bin/bash
# NOTE: needs tools/experiments/conv3x3_wide_chunks.hip copied over csrc/conv3x3.hip (shelved, DESIGN finding 37)
# GPU box: wide K-chunks (two 64-byte channel chunks staged per loop iteration) for the small-map conv3x3 launches
cd $GRAFT_REPO_ROOT
run() {
  echo "== $*"
  env "$@" python bench.py --no-cpu-baseline --no-other-dtype --steps 400 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('  ms/step', round(d['ms_per_step'],4), 'solo-sum', r['step_ms_sum_of_kernels'], 'in-situ-sum', r.get('in_situ',{}).get('step_ms_sum_of_kernels'))
for k,v in r['families'].items():
    if 'conv3x3' in k: print('   solo  ', k, v['launches_per_step'], v['avg_us'], v['ms_per_step'])
for k,v in r.get('in_situ',{}).get('families',{}).items():
    if 'conv3x3' in k: print('   insitu', k, v['launches_per_step'], v['avg_us'], v['ms_per_step'])
"
}
run LD_CONV_WIDE_MAX_WGS=0
run LD_CONV_WIDE_MAX_WGS=512
run LD_CONV_WIDE_MAX_WGS=256
run LD_CONV_WIDE_MAX_WGS=1024
run LD_CONV_WIDE_MAX_WGS=0
run LD_CONV_WIDE_MAX_WGS=512
