#!/bin/bash
# GPU box: the LDS-DMA ring small-map convolution (tools/experiments/conv3x3_ring.hip, finding 58) against the generic
# kernel: correctness, per launch alone on the chip, cycle stamps, and over the bench step (same-box alternating runs).
# Needs tools/ab/libdbg.so (tools/ab/build_dbg.sh).   usage: bash tools/exp_ring.sh ["variant env" ...]
cd $GRAFT_REPO_ROOT
export LD_LIB_OVERRIDE=$GRAFT_REPO_ROOT/tools/ab/libdbg.so
if [ $# = 0 ]; then set -- "LD_CONV_RING_R=2" "LD_CONV_RING_R=3" "LD_CONV_RING_IL=1"; fi
AB=("LD_CONV_RING=0")
for v in "$@"; do
  echo "#### $v"
  env $v python tools/experiments/trace_ring.py 2>&1 | grep -v "^$\|amdgpu.ids"
  env $v LD_CONV_RING_TRACE=1 python tools/experiments/trace_ring.py 2>&1 | grep -v "^$\|amdgpu.ids"
  AB+=("LD_CONV_RING=1 $v")
done
bash tools/ab/ab_env.sh "${AB[@]}"
