#!/bin/bash
# GPU box: pixel-chunk size of the fused linear-attention context kernel (LD_LINATTN_CHUNK_PX) -- per-launch view
cd $GRAFT_REPO_ROOT
run() {
  echo "== $*"
  env "$@" LD_BENCH_OPS=/tmp/ops.txt python bench.py --no-cpu-baseline --no-other-dtype --steps 300 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('  ms/step', round(d['ms_per_step'],4), 'solo-sum', r['step_ms_sum_of_kernels'], 'in-situ-sum', r.get('in_situ',{}).get('step_ms_sum_of_kernels'))
for k in ('linattn_kvctx','linattn_ctxfold','linattn_out'):
    print('   solo', k, r['families'][k]['avg_us'], ' in situ', r['in_situ']['families'][k]['avg_us'])
"
  grep -E "kvctx|ctxfold" /tmp/ops.txt | cut -c1-30,90-104 | tr '\n' ';'; echo
}
run LD_X=0
run LD_LINATTN_CHUNK_PX=256
run LD_LINATTN_CHUNK_PX=128
run LD_LINATTN_CHUNK_PX=64
