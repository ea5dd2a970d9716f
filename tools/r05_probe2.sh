#!/bin/bash
# round 5, GPU call 2: the lean Cout = 32 kernel -- parity, then alone on the chip against the generic kernel, then in the sampler
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/p2
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_hip_ops.py -q -x -k "conv3x3 or attention" > gpurun_out/p2/tests.txt 2>&1
SH="4,32,32,256,256;4,64,32,256,256;8,32,32,256,256"
for s32 in 0 1; do
  LD_CONV_S32=$s32 LD_CONV_C32=0 LD_BENCH_PRO=1 LD_BENCH_SHAPES="$SH" python tools/bench_conv.py > gpurun_out/p2/bench_conv_s32_$s32.txt 2>&1
done
for i in 1 2; do
  for s32 in 0 1; do
    LD_CONV_S32=$s32 python bench.py --no-cpu-baseline --no-other-dtype --no-roofline --steps 400 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('s32=$s32', round(d['ms_per_step'],4))" >> gpurun_out/p2/step_ab.txt
  done
done
LD_CONV_S32=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-dtype > gpurun_out/p2/bench20.json 2> gpurun_out/p2/bench20.err
echo done
