#!/bin/bash
# GPU box: N alternating rounds of arbitrary settings, each "<label>|<env assignments>" (LD_LIB_OVERRIDE=rel/path.so is made absolute);
# median per label.  usage: bash tools/ab/ab_cmd_n.sh <rounds> <steps> "off|LD_NO_SIDE_RES_CONV=1" "old|LD_LIB_OVERRIDE=tools/ab/libold.so LD_X=1" ...
cd $GRAFT_REPO_ROOT
N=$1; K=$2; shift 2
rm -f /tmp/ab_cmd_n.txt
for i in $(seq 1 $N); do
  for item in "$@"; do
    label=${item%%|*}; envs=${item#*|}
    envs=${envs//LD_LIB_OVERRIDE=tools/LD_LIB_OVERRIDE=$GRAFT_REPO_ROOT\/tools}
    env $envs python bench.py --no-cpu-baseline --no-other-dtype --no-roofline --no-legs --steps $K 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%-24s' % '$label', round(d['ms_per_step'],4))" | tee -a /tmp/ab_cmd_n.txt
  done
done
python - <<'PY'
import collections, statistics
acc = collections.OrderedDict()
for l in open('/tmp/ab_cmd_n.txt'):
    k, v = l.rsplit(None, 1)
    acc.setdefault(k.strip(), []).append(float(v))
for k, v in acc.items():
    print('median %-24s %.4f   (min %.4f max %.4f, n=%d)' % (k, statistics.median(v), min(v), max(v), len(v)))
PY
