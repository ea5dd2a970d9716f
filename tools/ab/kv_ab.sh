# GPU box: the working tree against tools/ab/libold.so (a build of another revision): linear-attention tests, then whole steps, alternating
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/kv
python -m pytest tests/test_hip_ops.py tests/test_hip_unet.py tests/test_hip_bench_shape.py -q -k "linear_attention or linattn or forward or bench_shape or chain" > gpurun_out/kv/tests.txt 2>&1; tail -3 gpurun_out/kv/tests.txt
bash tools/ab/lib_ab.sh libold.so ${1:-3} 400 > gpurun_out/kv/ab.txt 2>&1; cat gpurun_out/kv/ab.txt
for i in 1 2; do for which in tree other; do
  if [ $which = other ]; then export LD_LIB_OVERRIDE=$GRAFT_REPO_ROOT/tools/ab/libold.so; else unset LD_LIB_OVERRIDE; fi
  python bench.py --patches 64 --steps 60 --no-cpu-baseline --no-other-dtype --no-roofline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('p64 $which', round(d['ms_per_step'],4))"
  python bench.py --workload cfg5 --no-cpu-baseline --no-other-dtype --no-roofline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('cfg5 $which', round(d['value'],3), round(d['ms_per_step'],4))"
  python bench.py --dtype fp16 --steps 200 --no-cpu-baseline --no-other-dtype --no-roofline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('fp16 $which', round(d['value'],3), round(d['ms_per_step'],4))"
done; done > gpurun_out/kv/ab2.txt 2>&1; cat gpurun_out/kv/ab2.txt
