#!/bin/bash
# GPU box: like ab_env.sh at the driver's run length (--steps 20 --warmup 5), 5 alternating rounds
cd $GRAFT_REPO_ROOT
for i in $(seq 1 ${AB_ROUNDS:-8}); do
  for setting in "$@"; do
    env $setting python bench.py --no-cpu-baseline --no-other-dtype --no-roofline --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%-40s' % '$setting', round(d['ms_per_step'],4))"
  done
done
