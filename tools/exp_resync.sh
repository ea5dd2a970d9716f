cd $GRAFT_REPO_ROOT
run() { env $1 python bench.py --steps $2 --warmup 5 --no-cpu-baseline --no-other-dtype --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-44s' % '$1', $2, round(d['ms_per_step'],4))"; }
for i in 1 2; do
for s in "LD_X=0" "LD_SUB_RESYNC=0" "LD_SUB_RESYNC=8" "LD_SUB_RESYNC=128"; do run "$s" 400; done
for s in "LD_X=0" "LD_SUB_RESYNC_EARLY=0" "LD_SUB_RESYNC_EARLY=3"; do run "$s" 20; done
done
