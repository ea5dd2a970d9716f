#!/bin/bash
# GPU box: like tree_ab.sh at the driver's run length (--steps 20 --warmup 5), ${AB_ROUNDS:-8} alternating rounds.
cd $GRAFT_REPO_ROOT
for i in $(seq 1 ${AB_ROUNDS:-8}); do
  for which in new old; do
    if [ $which = old ]; then B=tools/ab/old_tree/bench.py; else B=bench.py; fi
    python $B --no-cpu-baseline --no-other-dtype --no-roofline --steps 20 --warmup 5 "$@" 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$which', round(d['ms_per_step'],4))"
  done
done
