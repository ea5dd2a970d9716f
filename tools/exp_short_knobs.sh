cd $GRAFT_REPO_ROOT
run() { env $1 python bench.py --steps $2 --warmup 5 --no-cpu-baseline --no-other-dtype --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-50s' % '$1', $2, round(d['ms_per_step'],4), round(d['ms_per_step']*$2,2))"; }
for i in 1 2; do
run "LD_X=0" 20
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" 20
run "LD_SUB_RESYNC=0" 20
run "LD_SUB_RESYNC_EARLY=0" 20
run "LD_SUB_RESYNC_EARLY=4" 20
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" 400
run "LD_X=0" 400
done
