#!/bin/bash
# GPU box: the shelved K-split small-map convolution (tools/experiments/conv3x3_ksplit.hip, finding 52) against the
# generic kernel, per launch (alone on the chip, launches back to back) and over the bench step (same-box alternating
# runs).  Needs a library built with csrc/build.sh --debug-variants:  LD_LIB_OVERRIDE=.../libdbg.so bash tools/exp_ksplit.sh
cd $GRAFT_REPO_ROOT
export LD_BENCH_SHAPES="4,256,256,32,32;8,256,256,32,32;4,128,128,32,32;4,128,256,32,32;4,64,64,64,64;4,128,128,64,64;4,256,128,64,64;4,64,64,128,128"
for s in 1 0; do
  LD_CONV_KSPLIT=$s LD_CONV_KSPLIT_MAX_PX=16384 LD_BENCH_PRO=1 python tools/bench_conv.py 2>&1 | grep -v "^$"
done
bash tools/ab/ab_env.sh "LD_CONV_KSPLIT=0" "LD_CONV_KSPLIT=1"
