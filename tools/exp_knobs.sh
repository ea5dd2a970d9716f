cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" python bench.py --no-cpu-baseline --no-roofline --steps 400 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  ', round(d['value'],3), round(d['ms_per_step'],4))"; }
run LD_X=0
run LD_NO_SEPARATE_ACT=1
run LD_CONV_SK=1
run LD_LINATTN_CHUNK_PX=256
run LD_LINATTN_CHUNK_PX=512
run LD_CONV_MT4_MIN_WGS=128
run LD_CONV_MT4_MIN_WGS=512
run LD_C1_SMALL_MIN=256
run LD_C1_SMALL_MIN=1024
run LD_CONV_BIG_MIN=256
run LD_CONV_BIG_MIN=1024
run LD_X=0
