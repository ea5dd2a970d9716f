#!/bin/bash
# GPU box: host cost of graph replay under the runtime's graph knobs (finding 40)
cd $GRAFT_REPO_ROOT
for setting in "LD_X=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "LD_SUB_BATCHES=4" "LD_SUB_BATCHES=4 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "LD_SUB_BATCHES=4 GPU_MAX_HW_QUEUES=8"; do
  echo "== $setting"
  env $setting python tools/exp_graph_host.py 400 2>&1 | grep -v amdgpu.ids
done
