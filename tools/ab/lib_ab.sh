cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for which in new old; do
    if [ $which = old ]; then export LD_LIB_OVERRIDE=$GRAFT_REPO_ROOT/tools/ab/libold.so; else unset LD_LIB_OVERRIDE; fi
    python bench.py --no-cpu-baseline --no-other-dtype --no-roofline --steps 600 "$@" 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$which', round(d['ms_per_step'],4))"
  done
done
