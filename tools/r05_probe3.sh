#!/bin/bash
# round 5, GPU call 3: the lean kernel with workgroup-level statistics -- parity, alone, and in the sampler per variant mask
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/p3
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_hip_ops.py -q -x -k "conv3x3" > gpurun_out/p3/tests.txt 2>&1
SH="4,32,32,256,256;4,64,32,256,256;8,32,32,256,256"
for s32 in 0 7; do
  LD_CONV_S32=$s32 LD_CONV_C32=0 LD_BENCH_PRO=1 LD_BENCH_SHAPES="$SH" python tools/bench_conv.py > gpurun_out/p3/bench_conv_s32_$s32.txt 2>&1
done
for i in 1 2 3; do
  for s32 in 0 1 3 7; do
    LD_CONV_S32=$s32 python bench.py --no-cpu-baseline --no-other-dtype --no-roofline --steps 400 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('s32=$s32', round(d['ms_per_step'],4))" >> gpurun_out/p3/step_ab.txt
  done
done
timeout 900 python -m pytest tests/test_hip_dist.py -q -x > gpurun_out/p3/tests_dist.txt 2>&1
echo done
