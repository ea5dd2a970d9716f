#!/bin/bash
# GPU box: the working tree's library against another build of it (tools/ab/<name>.so: build_plain.sh [flag] [name]),
# alternating runs of the default bench.  usage: bash tools/ab/lib_ab.sh <name.so> [rounds] [steps]
cd $GRAFT_REPO_ROOT
OTHER=$GRAFT_REPO_ROOT/tools/ab/${1:-libplain.so}
N=${2:-3}
K=${3:-600}
for i in $(seq 1 $N); do
  for which in tree other; do
    if [ $which = other ]; then export LD_LIB_OVERRIDE=$OTHER; else unset LD_LIB_OVERRIDE; fi
    python bench.py --no-cpu-baseline --no-other-dtype --no-roofline --no-legs --steps $K 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$which (${1:-libplain.so})', round(d['ms_per_step'],4))"
  done
done
