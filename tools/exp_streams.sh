#!/bin/bash
# GPU box: number of concurrent sub-batches x hardware queues (bench.py, cfg3).  (Round 2 also varied the number of
# host threads enqueueing the replays; it changed nothing, finding 29, and that switch is gone.)
cd $GRAFT_REPO_ROOT
for cfg in "2 2 4" "4 2 8" "4 2 4" "8 1 8" "3 2 8"; do
  set -- $cfg
  echo "== LD_SUB_BATCHES=$1 LD_MIN_SUB_BATCH=$2 GPU_MAX_HW_QUEUES=$3"
  LD_SUB_BATCHES=$1 LD_MIN_SUB_BATCH=$2 GPU_MAX_HW_QUEUES=$3 python bench.py --no-cpu-baseline --no-roofline --steps 300 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['concurrent_sub_batches'])"
done
