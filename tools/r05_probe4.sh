#!/bin/bash
# round 5, GPU call 4: K-mask halves debug; one forked graph per step vs one graph per stream; in-situ per-launch tables with / without the lean kernel
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/p4
export PYTHONUNBUFFERED=1
python tools/dbg_kmask.py > gpurun_out/p4/dbg_kmask.txt 2>&1
for i in 1 2 3; do
  for j in 0 1; do
    for k in 20 400; do
      w=5; [ $k = 400 ] && w=20
      LD_SUB_JOINT_GRAPH=$j python bench.py --no-cpu-baseline --no-other-dtype --no-roofline --steps $k --warmup $w 2>>gpurun_out/p4/joint.err | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('joint=$j steps=$k', round(d['ms_per_step'],4))" >> gpurun_out/p4/joint_ab.txt
    done
  done
done
for s32 in 0 3; do
  LD_CONV_S32=$s32 python bench.py --no-cpu-baseline --no-other-dtype --steps 400 > gpurun_out/p4/bench_s32_$s32.json 2> gpurun_out/p4/bench_s32_$s32.err
done
echo done
