cd $GRAFT_REPO_ROOT
run() { env $1 python bench.py --steps $2 --warmup 5 --no-cpu-baseline --no-other-dtype --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), end=' ')"; }
for s in "LD_X=0" "LD_SUB_AHEAD=0" "LD_SUB_AHEAD=2"; do
  echo; echo -n "$s 400: "; for i in 1 2 3 4 5 6; do run "$s" 400; done
  echo; echo -n "$s 20: "; for i in 1 2 3 4 5 6 7 8; do run "$s" 20; done
done
echo
