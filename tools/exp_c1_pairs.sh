cd $GRAFT_REPO_ROOT
bash tools/ab/ab_env.sh "LD_X=0" "LD_C1_PAIR_MAX_PX=262144" "LD_C1_PAIR_MAX_PX=1000000000" "LD_C1_PAIR_MAX_PX=1000000000 LD_C1_GROUP_MAX_PX=65536"
run() { env $1 python bench.py $2 --no-cpu-baseline --no-other-dtype --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-40s %s' % ('$1', '$2'), round(d['ms_per_step'],4))"; }
for i in 1 2; do for s in "LD_X=0" "LD_C1_PAIR_MAX_PX=262144" "LD_C1_PAIR_MAX_PX=1000000000"; do run "$s" "--patches 64 --steps 40"; LD_SUB_BATCHES=1 run "$s" "--steps 200"; run "$s" "--workload cfg5"; done; done
