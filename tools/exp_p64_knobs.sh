cd $GRAFT_REPO_ROOT
run() { env $1 python bench.py --patches 64 --steps 40 --no-cpu-baseline --no-other-dtype --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-40s' % '$1', round(d['ms_per_step'],4), round(d['value'],3))"; }
for i in 1 2; do for s in "LD_X=0" "LD_CONV_NO_C32=1" "LD_CONV_C32_MIN_TILES=16384" "LD_CONV_C32_R=4" "LD_SUB_BATCHES=1" "LD_SUB_BATCHES=4" "LD_CONV_BIG_MIN=4096" "LD_LINATTN_CHUNK_PX=512,256,128"; do run "$s"; done; done
