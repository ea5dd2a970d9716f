cd $GRAFT_REPO_ROOT
export LD_LIB_OVERRIDE=$GRAFT_REPO_ROOT/tools/ab/libdbg.so
run() { env $1 python bench.py --no-cpu-baseline --no-other-dtype --no-roofline $2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%-50s' % '$1 $2', round(d['ms_per_step'],4), round(d['value'],3))"; }
for i in 1 2; do
for s in "LD_CONV_RING=0" "LD_CONV_RING=1"; do
  run "$s" "--patches 64 --steps 60"
  run "$s LD_SUB_BATCHES=1" "--steps 300"
  run "$s" "--patches 16 --steps 200"
  run "$s" "--patches 32 --steps 100"
done; done
