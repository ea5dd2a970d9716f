#!/bin/bash
# GPU box (library built with --debug-variants): ablation of the generic conv3x3 on the small-map shapes.
# LD_CONV_DEBUG bits: 1 no halo loads, 2 no weight loads, 4 no MFMA (compute() skipped, incl. its LDS fragment reads), 8 no stores
cd $GRAFT_REPO_ROOT
bash localdiffusion-hallucination_amd/csrc/build.sh --debug-variants 2>&1 | tail -1
for dbg in 0 1 2 3 4 7 8 12 15; do
  LD_CONV_DEBUG=$dbg LD_BENCH_SEL=5,7 LD_CONV_NO_C32=1 python tools/bench_conv.py 2>&1 | grep "stats=True"
done
