#!/bin/bash
# GPU box: alternate the bench of tools/ab/old_tree (tools/ab/export_tree.sh <rev>) and of the working tree, 3 rounds.
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for which in new old; do
    if [ $which = old ]; then B=tools/ab/old_tree/bench.py; else B=bench.py; fi
    python $B --no-cpu-baseline --no-other-dtype --no-roofline --steps 600 "$@" 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$which', round(d['ms_per_step'],4))"
  done
done
