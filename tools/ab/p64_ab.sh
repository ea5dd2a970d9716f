cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/wt
for i in 1 2; do
for s in "LD_CONV_C32=1" "LD_CONV_C32=0" "LD_CONV_C32=0 LD_CONV_S32=7"; do
  env $s python bench.py --patches 64 --steps 60 --no-cpu-baseline --no-other-dtype --no-roofline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('p64 %-30s' % '$s', round(d['value'],3), round(d['ms_per_step'],4))" >> gpurun_out/wt/p64_ab.txt
  env $s LD_SUB_BATCHES=1 python bench.py --steps 200 --no-cpu-baseline --no-other-dtype --no-roofline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('solo8 %-30s' % '$s', round(d['value'],3), round(d['ms_per_step'],4))" >> gpurun_out/wt/p64_ab.txt
done; done
SH="8,32,32,256,256;32,32,32,256,256"
LD_CONV_C32=1 LD_BENCH_PRO=1 LD_BENCH_SHAPES="$SH" python tools/bench_conv.py > gpurun_out/wt/conv_c32.txt 2>&1
LD_CONV_C32=0 LD_BENCH_PRO=1 LD_BENCH_SHAPES="$SH" python tools/bench_conv.py > gpurun_out/wt/conv_s32.txt 2>&1
