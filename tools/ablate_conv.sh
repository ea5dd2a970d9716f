#!/bin/bash
# GPU box: true kernel durations (rocprofv3 kernel trace) of ld_conv3x3 under the LD_CONV_DEBUG ablation bits.
# usage: ABLATE_BITS="0 1 4" LD_BENCH_SEL=0,1 LD_BENCH_PRO=1 bash tools/ablate_conv.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/ablate
rm -rf $OUT; mkdir -p $OUT
for dbg in ${ABLATE_BITS:-0 1 2 3 4 7 8 15}; do
  export LD_CONV_DEBUG=$dbg
  timeout 120 rocprofv3 --kernel-trace --output-format csv -d $OUT/d$dbg -o r -- python3 $R/tools/bench_conv.py > $OUT/log_$dbg.txt 2>&1 < /dev/null
done
python3 $R/tools/ablate_summary.py $OUT
