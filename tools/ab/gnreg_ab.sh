#!/bin/bash
# GPU box: same-box A/B of the register GroupNorm coefficients (library table entry gn_reg_coef, LD_GN_REG_COEF):
#   0 = LDS chain everywhere, 1 = gn_apply (fragment inside one group), 5 = + gn_apply at C = 32, 3 / 7 = + conv1x1's GroupNorm tails.
# usage: bash tools/ab/gnreg_ab.sh "0 1 5 7" [steps]   (the kernels of this experiment: git show eea172d; measured slower, reverted -- docs/findings.md 111)
cd $GRAFT_REPO_ROOT
SET=${1:-"0 1 5"}
STEPS=${2:-400}
for i in 1 2 3; do
  for m in $SET; do
    LD_GN_REG_COEF=$m python bench.py --no-cpu-baseline --no-other-dtype --no-roofline --no-legs --steps $STEPS 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('gn_reg_coef=%-3s' % '$m', round(d['ms_per_step'],4))"
  done
done
for m in $SET; do
  LD_GN_REG_COEF=$m LD_BENCH_OPS=/tmp/ops_$m.txt python bench.py --no-cpu-baseline --no-other-dtype --no-legs --steps 100 > /dev/null 2>&1
  python - <<PY
import re
rows=[l for l in open('/tmp/ops_$m.txt')]
i=rows.index('# in_situ\n') if '# in_situ\n' in rows else 0
gn=[float(re.search(r'([0-9.]+) us',l).group(1)) for l in rows[i:] if ' gn_apply ' in l]
ct=[float(re.search(r'([0-9.]+) us',l).group(1)) for l in rows[i:] if 'res_conv+tail' in l]
print('gn_reg_coef=$m in situ: gn_apply %d launches avg %.2f us (sum %.1f); res_conv+tail %d launches avg %.2f us (sum %.1f)' % (len(gn), sum(gn)/max(len(gn),1), sum(gn), len(ct), sum(ct)/max(len(ct),1), sum(ct)))
PY
done
