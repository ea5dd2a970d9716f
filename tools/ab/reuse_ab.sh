# GPU box: Tuning.buffer_reuse (activations of a sampler plan in one pool by liveness) against a buffer per layer, alternating
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/reuse; : > gpurun_out/reuse/ab.txt
for i in 1 2 3; do
for v in 0 1; do
  LD_BUFFER_REUSE=$v python bench.py --steps 400 --no-cpu-baseline --no-other-dtype --no-roofline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('p8   reuse=$v', round(d['value'],3), round(d['ms_per_step'],4))" >> gpurun_out/reuse/ab.txt
  LD_BUFFER_REUSE=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-dtype --no-roofline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('p8s20 reuse=$v', round(d['value'],3), round(d['ms_per_step'],4))" >> gpurun_out/reuse/ab.txt
done; done
for i in 1 2; do for v in 0 1; do
  LD_BUFFER_REUSE=$v python bench.py --patches 64 --steps 60 --no-cpu-baseline --no-other-dtype --no-roofline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('p64  reuse=$v', round(d['value'],3), round(d['ms_per_step'],4))" >> gpurun_out/reuse/ab.txt
  LD_BUFFER_REUSE=$v python bench.py --workload cfg5 --no-cpu-baseline --no-other-dtype --no-roofline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('cfg5 reuse=$v', round(d['value'],3), round(d['ms_per_step'],4))" >> gpurun_out/reuse/ab.txt
  LD_BUFFER_REUSE=$v LD_SUB_BATCHES=1 python bench.py --steps 200 --no-cpu-baseline --no-other-dtype --no-roofline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('solo8 reuse=$v', round(d['value'],3), round(d['ms_per_step'],4))" >> gpurun_out/reuse/ab.txt
done; done
cat gpurun_out/reuse/ab.txt
