#!/bin/bash
# GPU box: the round-3 experiment records that go with tools/profile_round.sh <tag>:
#   <tag>_xcd_barrier_probe.txt   tools/probes/xcd_barrier (kernel boundary vs agent-scope vs XCD-local phase boundaries)
#   <tag>_stage_trace.txt         stage programs: per-launch durations next to the stand-alone launches + per-phase cycle stamps
#   <tag>_stage_ab.txt            same-box A/B of the bench step with / without stage programs
#   <tag>_ksplit.txt              the shelved K-split small-map convolution: per-launch times, cycle stamps, step A/B
#   <tag>_halfchip_c32.txt        persistent C=32 convolution on half the chip under XCD-striped CU masks: step A/B
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
cd $R
timeout 120 tools/probes/xcd_barrier 200 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_xcd_barrier_probe.txt
timeout 300 python3 tools/trace_stage.py 4 256 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_stage_trace.txt
timeout 600 bash tools/ab/ab_env.sh "LD_STAGE_MAX_PX=0" "LD_STAGE_MAX_PX=1024" 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_stage_ab.txt
if [ -f tools/ab/libdbg.so ]; then
  export LD_LIB_OVERRIDE=$R/tools/ab/libdbg.so
  ( timeout 300 python3 tools/experiments/trace_ksplit.py; timeout 900 bash tools/exp_ksplit.sh ) 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_ksplit.txt
  unset LD_LIB_OVERRIDE
fi
timeout 900 bash tools/exp_halfchip.sh 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_halfchip_c32.txt
tail -3 $OUT/${TAG}_stage_ab.txt
