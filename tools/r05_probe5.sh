#!/bin/bash
# round 5, GPU call 5: correctness of the round's new paths (p_sample row mode, K-mask halves, G16 pins, attention mapping),
# attention alone with / without the XCD mapping, cfg5 and cfg3 with / without it
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/p5
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_hip_dist.py tests/test_hip_sampler.py -q -x -k "kmask or p_sample or sharded or shard" > gpurun_out/p5/tests_a.txt 2>&1
timeout 900 python -m pytest tests/test_hip_ops.py -q -x -k "attention" > gpurun_out/p5/tests_b.txt 2>&1
timeout 1200 python -m pytest tests/test_hip_lowp_chain.py -q -s -k "contractive" > gpurun_out/p5/tests_g16.txt 2>&1
for m in 0 1; do
  LD_ATTN_XCD_MAP=$m python tools/bench_attention.py > gpurun_out/p5/attn_map_$m.txt 2>&1
done
for i in 1 2 3; do
  for m in 0 1; do
    LD_ATTN_XCD_MAP=$m python bench.py --workload cfg5 --no-roofline --steps 500 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('cfg5 map=$m', round(d['value'],3), round(d['ms_per_step'],4))" >> gpurun_out/p5/cfg5_ab.txt
    LD_ATTN_XCD_MAP=$m python bench.py --no-cpu-baseline --no-other-dtype --no-roofline --steps 400 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('cfg3 map=$m', round(d['ms_per_step'],4))" >> gpurun_out/p5/cfg3_ab.txt
  done
done
echo done
