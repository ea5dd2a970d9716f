#!/bin/bash
cd $GRAFT_REPO_ROOT
for setting in "LD_X=0" "GPU_MAX_HW_QUEUES=2" "GPU_MAX_HW_QUEUES=8" "GPU_MAX_HW_QUEUES=16" "DEBUG_HIP_DYNAMIC_QUEUES=0" "DEBUG_HIP_DYNAMIC_QUEUES=1" "DEBUG_HIP_FORCE_GRAPH_QUEUES=4" "DEBUG_HIP_FORCE_ASYNC_QUEUE=1"; do
  echo "== $setting"; env $setting timeout 120 tools/probes/queue_concurrency 0
done
echo "== priorities"; timeout 120 tools/probes/queue_concurrency 1
echo "== priorities GPU_MAX_HW_QUEUES=8"; GPU_MAX_HW_QUEUES=8 timeout 120 tools/probes/queue_concurrency 1
