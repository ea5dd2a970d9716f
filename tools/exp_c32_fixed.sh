#!/bin/bash
# GPU box: where the persistent C = 32 convolution's fixed cost goes.  Cycle stamps of workgroup
# (0,0) (library built by tools/ab/build_dbg.sh) and event-timed launches back to back, for 4 / 8 / 16 / 32 patches.
cd $GRAFT_REPO_ROOT
export LD_LIB_OVERRIDE=$GRAFT_REPO_ROOT/tools/ab/libdbg.so LD_CONV_C32_MIN_TILES=1024
LD_TRACE_B=4,8,16,32 python tools/trace_c32.py 2>&1 | grep -v "^$"
export LD_BENCH_SHAPES="4,32,32,256,256;8,32,32,256,256;16,32,32,256,256;32,32,32,256,256"
LD_BENCH_PRO=1 python tools/bench_conv.py 2>&1 | grep -v "^$"
for d in 1 4 8 13 45; do LD_CONV_DEBUG=$d LD_BENCH_PRO=1 LD_BENCH_SHAPES="4,32,32,256,256;8,32,32,256,256" python tools/bench_conv.py 2>&1 | grep -v "^$"; done
