# GPU box: the 64-channel x 16-row conv3x3 tile (<4,4>, conv_big4_min) against the default routing, alternating, same box.
# usage: bash tools/ab/big4_ab.sh [rounds]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/big4
N=${1:-2}
python -m pytest tests/test_hip_ops.py -q -k "big64 or conv3x3" > gpurun_out/big4/tests.txt 2>&1; tail -1 gpurun_out/big4/tests.txt
: > gpurun_out/big4/ab.txt
for i in $(seq $N); do
for v in off 2048 1024 512 256; do
  if [ $v = off ]; then E="LD_UNUSED=1"; else E="LD_CONV_BIG4_MIN=$v"; fi
  env $E python bench.py --patches 64 --steps 60 --no-cpu-baseline --no-other-dtype --no-roofline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('p64   %-6s' % '$v', round(d['value'],3), round(d['ms_per_step'],4))" >> gpurun_out/big4/ab.txt
  env $E LD_SUB_BATCHES=1 python bench.py --patches 64 --steps 60 --no-cpu-baseline --no-other-dtype --no-roofline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('p64s1 %-6s' % '$v', round(d['value'],3), round(d['ms_per_step'],4))" >> gpurun_out/big4/ab.txt
  env $E python bench.py --steps 200 --no-cpu-baseline --no-other-dtype --no-roofline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('p8    %-6s' % '$v', round(d['value'],3), round(d['ms_per_step'],4))" >> gpurun_out/big4/ab.txt
done; done
cat gpurun_out/big4/ab.txt
