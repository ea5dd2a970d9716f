# GPU box: kvctx chunk sizes again after finding 105 (LD_LINATTN_CHUNK_PX = px per chunk at n >= 65536 / >= 16384 / smaller), alternating
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/chunk; : > gpurun_out/chunk/ab.txt
for i in 1 2 3; do
for v in "512,256,128" "256,256,128" "1024,256,128" "512,128,128" "512,512,128" "512,256,64" "512,256,256" "256,128,64"; do
  env LD_LINATTN_CHUNK_PX=$v python bench.py --steps 400 --no-cpu-baseline --no-other-dtype --no-roofline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('p8  chunk=%-16s' % '$v', round(d['value'],3), round(d['ms_per_step'],4))" >> gpurun_out/chunk/ab.txt
done; done
cat gpurun_out/chunk/ab.txt
