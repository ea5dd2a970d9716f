#!/bin/bash
# GPU box: the persistent C = 32 convolution on a 256-thread / 48-KB footprint ("lite", finding 66) for the 4-patch launches
# of the two-sub-batch regime: correctness (the op tests with every eligible launch on it), per launch alone, step A/B.
cd $GRAFT_REPO_ROOT
LD_CONV_C32_LITE=1 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "conv3x3" 2>&1 | grep -E "passed|failed|Error" | tail -3
export LD_BENCH_SHAPES="4,32,32,256,256;8,32,32,256,256;4,32,32,128,128"
for l in 0 1024; do LD_CONV_C32_LITE=$l LD_BENCH_PRO=1 python tools/bench_conv.py 2>&1 | grep -v "^$\|amdgpu.ids"; done
bash tools/ab/ab_env.sh "LD_CONV_C32_LITE=0" "LD_CONV_C32_LITE=1024" "LD_CONV_C32_LITE=1024 LD_CONV_C32_LITE_R=6" "LD_CONV_C32_LITE=1024 LD_CONV_C32_LITE_WGS=512" "LD_CONV_C32_LITE=256"
