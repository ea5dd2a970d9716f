#!/bin/bash
# round 5, GPU call 1: baselines on this box -- short / long bench lines, the short run's fixed part, the conv-path
# launches alone (plain / statistics / prologue) and the cycle stamps of one of their workgroups
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/p1
export PYTHONUNBUFFERED=1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-dtype > gpurun_out/p1/bench20.json 2> gpurun_out/p1/bench20.err
python bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-other-dtype --no-roofline > gpurun_out/p1/bench400.json 2> gpurun_out/p1/bench400.err
python tools/fixed_part.py > gpurun_out/p1/fixed_part.txt 2>&1
LD_PREHEAT=300 python tools/fixed_part.py > gpurun_out/p1/fixed_part_preheat.txt 2>&1
python tools/timeline_short_run.py 20 > gpurun_out/p1/timeline20.txt 2>&1
export LD_LIB_OVERRIDE=$GRAFT_REPO_ROOT/tools/ab/libdbg.so
LD_CONV_NO_C32=1 LD_BENCH_PRO=1 LD_BENCH_SHAPES="4,32,32,256,256;4,64,32,256,256;8,32,32,256,256;4,256,256,32,32;4,128,128,64,64" python tools/bench_conv.py > gpurun_out/p1/bench_conv.txt 2>&1
LD_TRACE_SHAPES="4,32,32,256,256,0;4,32,32,256,256,1;4,64,32,256,256,0;4,256,256,32,32,0;4,256,256,32,32,1" python tools/trace_conv.py > gpurun_out/p1/trace_conv.txt 2>&1
echo done
