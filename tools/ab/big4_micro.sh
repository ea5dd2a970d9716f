# GPU box: per-launch times of the 64-channel conv3x3 tiles (<4,2> against <4,4>) at the throughput regime's shapes
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/big4
SH="64,256,256,32,32;64,128,128,64,64;64,64,64,128,128;32,256,256,32,32;32,128,128,64,64;32,64,64,128,128;64,512,256,32,32;4,64,64,128,128;4,128,128,64,64;4,256,256,32,32"
for v in off 1; do
  if [ $v = off ]; then E="LD_UNUSED=1"; else E="LD_CONV_BIG4_MIN=$v"; fi
  env $E LD_BENCH_PRO=1 LD_BENCH_SHAPES="$SH" python tools/bench_conv.py > gpurun_out/big4/micro_$v.txt 2>&1
done
paste -d'|' gpurun_out/big4/micro_off.txt gpurun_out/big4/micro_1.txt | awk -F'|' '{print $1; print "   <4,4>:" substr($2, index($2, ":")+1)}' > gpurun_out/big4/micro.txt
cat gpurun_out/big4/micro.txt
: > gpurun_out/big4/ab2.txt
for i in 1 2; do
for v in off 256 128 64 1; do
  if [ $v = off ]; then E="LD_UNUSED=1"; else E="LD_CONV_BIG4_MIN=$v"; fi
  env $E python bench.py --steps 200 --no-cpu-baseline --no-other-dtype --no-roofline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('p8    %-6s' % '$v', round(d['value'],3), round(d['ms_per_step'],4))" >> gpurun_out/big4/ab2.txt
  env $E python bench.py --patches 64 --steps 60 --no-cpu-baseline --no-other-dtype --no-roofline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('p64   %-6s' % '$v', round(d['value'],3), round(d['ms_per_step'],4))" >> gpurun_out/big4/ab2.txt
done; done
cat gpurun_out/big4/ab2.txt
