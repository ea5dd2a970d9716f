#!/bin/bash
# GPU box: the persistent C=32 conv in the solo (B=8) and the two-sub-batch (B=4 per launch) regime, ring depth / grid width sweeps
cd $GRAFT_REPO_ROOT
run() {
  echo "== $*"
  env "$@" python bench.py --no-cpu-baseline --steps 300 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('  ms/step', round(d['ms_per_step'],4), 'solo-sum', r['step_ms_sum_of_kernels'], 'rb solo', round(r['resblock_conv_path']['ms_per_step'],4), round(r['resblock_conv_path']['hbm_frac'],3),
      'in-situ', r.get('in_situ',{}).get('step_ms_sum_of_kernels'), r.get('in_situ',{}).get('resblock_conv_path',{}).get('ms_per_step'))
for k,v in r['families'].items():
    if 'c32' in k or '2,4' in k: print('   solo', k, v['launches_per_step'], v['avg_us'])
for k,v in r.get('in_situ',{}).get('families',{}).items():
    if 'c32' in k or '2,4' in k: print('   insitu', k, v['launches_per_step'], v['avg_us'])
"
}
run LD_X=0
run LD_CONV_C32_R=4
run LD_CONV_C32_MIN_TILES=1024
run LD_CONV_C32_MIN_TILES=1024 LD_CONV_C32_R=4
run LD_CONV_C32_MIN_TILES=1024 LD_CONV_C32_R=4 LD_CONV_C32_CUS=128
run LD_CONV_C32_MIN_TILES=1024 LD_CONV_C32_CUS=128
run LD_CONV_NO_C32=1
