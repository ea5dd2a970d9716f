#!/bin/bash
# GPU box: the working tree's library against several other builds (tools/ab/<name>.so ...), alternating runs.
# usage: bash tools/ab/multi_ab.sh <rounds> <steps> name1.so name2.so ...
cd $GRAFT_REPO_ROOT
N=$1; K=$2; shift 2
for i in $(seq 1 $N); do
  for which in tree "$@"; do
    if [ $which = tree ]; then unset LD_LIB_OVERRIDE; else export LD_LIB_OVERRIDE=$GRAFT_REPO_ROOT/tools/ab/$which; fi
    python bench.py --no-cpu-baseline --no-other-dtype --no-roofline --no-legs --steps $K 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%-12s' % '$which', round(d['ms_per_step'],4))"
  done
done
