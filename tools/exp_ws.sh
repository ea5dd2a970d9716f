#!/bin/bash
# GPU box: the small-map weights-in-registers kernel (conv3x3_ws.hip) against the generic kernel, per launch (alone on
# the chip, launches back to back) and over the bench step (same-box alternating runs).  usage: bash tools/exp_ws.sh
cd $GRAFT_REPO_ROOT
export LD_BENCH_SHAPES="4,256,256,32,32;8,256,256,32,32;4,128,128,32,32;4,128,256,32,32;4,64,64,64,64;4,128,128,64,64;4,256,128,64,64;4,64,64,128,128"
for s in 0 1; do
  LD_CONV_NO_WS=$s LD_CONV_WS_MAX_PX=16384 LD_BENCH_PRO=1 python tools/bench_conv.py 2>&1 | grep -v "^$"
done
bash tools/ab/ab_env.sh "LD_X=0" "LD_CONV_NO_WS=1"
