#!/bin/bash
# GPU box: same-box A/B of environment settings, N alternating rounds, median per setting at the end.
# usage: bash tools/ab/ab_env_n.sh <rounds> <steps> "LD_X=0" "LD_Y=1" ...
cd $GRAFT_REPO_ROOT
N=$1; K=$2; shift 2
rm -f /tmp/ab_env_n.txt
for i in $(seq 1 $N); do
  for setting in "$@"; do
    env $setting python bench.py --no-cpu-baseline --no-other-dtype --no-roofline --no-legs --steps $K 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%-44s' % '$setting', round(d['ms_per_step'],4))" | tee -a /tmp/ab_env_n.txt
  done
done
python - <<'PY'
import collections, statistics
acc = collections.OrderedDict()
for l in open('/tmp/ab_env_n.txt'):
    k, v = l.rsplit(None, 1)
    acc.setdefault(k.strip(), []).append(float(v))
for k, v in acc.items():
    print('median %-44s %.4f   (min %.4f max %.4f, n=%d)' % (k, statistics.median(v), min(v), max(v), len(v)))
PY
