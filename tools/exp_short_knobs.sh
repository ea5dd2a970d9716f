#!/bin/bash
# GPU box: does anything shorten the slow first ~40 steps of a run (the driver's --steps 20)?  ms per step, total ms.
cd $GRAFT_REPO_ROOT
run() { env $1 python bench.py --steps $2 --warmup 5 --no-cpu-baseline --no-other-dtype --no-roofline $3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-60s' % '$1 $3', $2, round(d['ms_per_step'],4), round(d['ms_per_step']*$2,2))"; }
for i in 1 2; do
for s in "LD_X=0" "LD_SUB_AHEAD=1" "LD_SUB_AHEAD=1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1"; do run "$s" 20; run "$s" 400; done
run "LD_X=0" 20 "--patches 64"
run "LD_SUB_AHEAD=1" 20 "--patches 64"
run "LD_X=0" 100 "--patches 16"
run "LD_SUB_AHEAD=1" 100 "--patches 16"
done
