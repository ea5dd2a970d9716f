#!/bin/bash
# NOTE: needs tools/experiments/conv3x3_fat_waves.hip copied over csrc/conv3x3.hip (the variant is shelved, DESIGN finding 32)
# GPU box: fat-wave conv3x3 variants (LD_CONV_FAT = 0 off / 1 the 32-channel-tile launches / 2 also the 64-channel-tile ones)
cd $GRAFT_REPO_ROOT
run() {
  echo "== $*"
  env "$@" python bench.py --no-cpu-baseline --steps 400 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('  ms/step', round(d['ms_per_step'],4), 'solo-sum', r['step_ms_sum_of_kernels'], 'in-situ-sum', r.get('in_situ',{}).get('step_ms_sum_of_kernels'))
for k,v in r['families'].items():
    if 'conv3x3' in k: print('   solo  ', k, v['launches_per_step'], v['avg_us'], v['ms_per_step'])
for k,v in r.get('in_situ',{}).get('families',{}).items():
    if 'conv3x3' in k: print('   insitu', k, v['launches_per_step'], v['avg_us'], v['ms_per_step'])
"
}
run LD_CONV_FAT=0
run LD_CONV_FAT=1
run LD_CONV_FAT=2
run LD_CONV_FAT=0
run LD_CONV_FAT=1
