#!/bin/bash
# GPU box: same-box A/B of environment settings (alternating runs, 3 rounds).  usage: bash tools/ab/ab_env.sh "LD_X=0" "LD_CONV_SK=1" ...
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for setting in "$@"; do
    env $setting python bench.py --no-cpu-baseline --no-other-dtype --no-roofline --steps 400 --no-legs 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%-40s' % '$setting', round(d['ms_per_step'],4))"
  done
done
