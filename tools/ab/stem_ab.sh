# GPU box: the stem kernel of the working tree against tools/ab/libold.so (HEAD): alone on the chip, then whole steps, alternating
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/stem
python -m pytest tests/test_hip_ops.py -q -k "stem or conv_image" > gpurun_out/stem/tests.txt 2>&1; tail -1 gpurun_out/stem/tests.txt
(echo "== tree"; python tools/bench_stem.py; echo "== old"; LD_LIB_OVERRIDE=$GRAFT_REPO_ROOT/tools/ab/libold.so python tools/bench_stem.py) 2>&1 | grep -v amdgpu.ids > gpurun_out/stem/micro.txt
cat gpurun_out/stem/micro.txt
bash tools/ab/lib_ab.sh libold.so ${1:-3} 400 > gpurun_out/stem/ab.txt 2>&1; cat gpurun_out/stem/ab.txt
for i in 1 2; do for which in tree other; do
  if [ $which = other ]; then export LD_LIB_OVERRIDE=$GRAFT_REPO_ROOT/tools/ab/libold.so; else unset LD_LIB_OVERRIDE; fi
  python bench.py --patches 64 --steps 60 --no-cpu-baseline --no-other-dtype --no-roofline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('p64 $which', round(d['ms_per_step'],4))"
  python bench.py --workload cfg5 --no-cpu-baseline --no-other-dtype --no-roofline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('cfg5 $which', round(d['value'],3), round(d['ms_per_step'],4))"
done; done > gpurun_out/stem/ab2.txt 2>&1; cat gpurun_out/stem/ab2.txt
