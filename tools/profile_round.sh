#!/bin/bash
# GPU box: the evidence set of one round, all from ONE box.  usage: bash tools/profile_round.sh r02_a
#   1. default bench line (two concurrent sub-batches)                    -> gpurun_out/<tag>_bench.json
#   2. the same with LD_SUB_BATCHES=1 (one batch, one stream: the regime `roofline.frac` prices)  -> <tag>_s1_bench.json
#   3. rocprofv3 --kernel-trace --stats of both commands (fewer steps)    -> <tag>_kernel_stats.csv, <tag>_s1_kernel_stats.csv
#   4. separate --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of the solo regime -> <tag>_s1_pmc_traffic.json / _pmc_summary.txt
#   5. the same passes in the default regime -> <tag>_pmc_traffic.json / _pmc_summary.txt;  6. cfg5 bench line + kernel trace + its own
#   --pmc passes (<tag>_cfg5_pmc_traffic.json);  7. the conv path's chip-level leg alone, un-profiled and as a kernel trace
# rocprofv3's interception slows the graph launches of the two-sub-batch regime (its kernels then overlap less than
# un-profiled); the single-stream eager run is hardly perturbed, which is why the roofline numbers are tied to it.
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $OUT/${TAG}_bench.log 2>&1 < /dev/null
tail -1 $OUT/${TAG}_bench.log > $OUT/${TAG}_bench.json
LD_SUB_BATCHES=1 python3 $R/bench.py --no-cpu-baseline --no-other-dtype > $OUT/${TAG}_s1_bench.log 2>&1 < /dev/null
tail -1 $OUT/${TAG}_s1_bench.log > $OUT/${TAG}_s1_bench.json
python3 $R/bench.py --dtype fp16 --no-cpu-baseline --no-other-dtype > $OUT/${TAG}_fp16_bench.log 2>&1 < /dev/null
tail -1 $OUT/${TAG}_fp16_bench.log > $OUT/${TAG}_fp16_bench.json
rm -rf /tmp/prof_ks /tmp/prof_ks1 /tmp/prof_f /tmp/prof_w
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ks -o r -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-roofline --no-other-dtype > $OUT/${TAG}_ks.log 2>&1 < /dev/null
cp $(find /tmp/prof_ks -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_kernel_stats.csv
python3 $R/tools/conv_path_union.py $(find /tmp/prof_ks -name '*kernel_trace.csv' | head -1) > $OUT/${TAG}_conv_path_union.txt 2>&1
LD_SUB_BATCHES=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ks1 -o r -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-roofline --no-other-dtype > $OUT/${TAG}_s1_ks.log 2>&1 < /dev/null
cp $(find /tmp/prof_ks1 -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_s1_kernel_stats.csv
LD_SUB_BATCHES=1 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_f -o r -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-other-dtype > $OUT/${TAG}_pf.log 2>&1 < /dev/null
LD_SUB_BATCHES=1 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_w -o r -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-other-dtype > $OUT/${TAG}_pw.log 2>&1 < /dev/null
python3 $R/tools/pmc_summarize.py /tmp/prof_f /tmp/prof_w $OUT/${TAG}_s1_pmc_traffic.json $OUT/${TAG}_s1_bench.json > $OUT/${TAG}_s1_pmc_summary.txt 2>&1
# 5. the same two --pmc passes in the DEFAULT regime (two concurrent sub-batches of 4: what `roofline.frac` prices and
#    `roofline.traffic` quotes; per launch of 4 patches)               -> <tag>_pmc_traffic.json / _pmc_summary.txt
rm -rf /tmp/prof_f2 /tmp/prof_w2
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_f2 -o r -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-other-dtype > $OUT/${TAG}_pf2.log 2>&1 < /dev/null
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_w2 -o r -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-other-dtype > $OUT/${TAG}_pw2.log 2>&1 < /dev/null
python3 $R/tools/pmc_summarize.py /tmp/prof_f2 /tmp/prof_w2 $OUT/${TAG}_pmc_traffic.json $OUT/${TAG}_bench.json > $OUT/${TAG}_pmc_summary.txt 2>&1
# 6. cfg5 (512^2, DDIM 50, branch + fusion, fp16): bench line with its roofline block + kernel trace
python3 $R/bench.py --workload cfg5 --no-cpu-baseline > $OUT/${TAG}_cfg5_bench.log 2>&1 < /dev/null
tail -1 $OUT/${TAG}_cfg5_bench.log > $OUT/${TAG}_cfg5_bench.json
rm -rf /tmp/prof_k5
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k5 -o r -- python3 $R/bench.py --workload cfg5 --steps 100 --no-cpu-baseline --no-roofline > $OUT/${TAG}_cfg5_ks.log 2>&1 < /dev/null
cp $(find /tmp/prof_k5 -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_cfg5_kernel_stats.csv
# 6b. cfg5's own --pmc passes (its dominant conv3x3<f16,2,2> had no traffic figure) -> <tag>_cfg5_pmc_traffic.json
rm -rf /tmp/prof_f5 /tmp/prof_w5
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_f5 -o r -- python3 $R/bench.py --workload cfg5 --steps 50 --no-cpu-baseline --no-roofline > $OUT/${TAG}_pf5.log 2>&1 < /dev/null
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_w5 -o r -- python3 $R/bench.py --workload cfg5 --steps 50 --no-cpu-baseline --no-roofline > $OUT/${TAG}_pw5.log 2>&1 < /dev/null
python3 $R/tools/pmc_summarize.py /tmp/prof_f5 /tmp/prof_w5 $OUT/${TAG}_cfg5_pmc_traffic.json $OUT/${TAG}_cfg5_bench.json > $OUT/${TAG}_cfg5_pmc_summary.txt 2>&1
# 6c. the same for cfg4's per-GPU share (64 patches per GPU) -> <tag>_p64_pmc_traffic.json (the default line's cfg4_share leg quotes it)
rm -rf /tmp/prof_f6 /tmp/prof_w6
python3 $R/bench.py --patches 64 --steps 20 --warmup 5 --no-cpu-baseline --no-other-dtype > $OUT/${TAG}_p64s_bench.log 2>&1 < /dev/null
tail -1 $OUT/${TAG}_p64s_bench.log > $OUT/${TAG}_p64s_bench.json
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_f6 -o r -- python3 $R/bench.py --patches 64 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-other-dtype > $OUT/${TAG}_pf6.log 2>&1 < /dev/null
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_w6 -o r -- python3 $R/bench.py --patches 64 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-other-dtype > $OUT/${TAG}_pw6.log 2>&1 < /dev/null
python3 $R/tools/pmc_summarize.py /tmp/prof_f6 /tmp/prof_w6 $OUT/${TAG}_p64_pmc_traffic.json $OUT/${TAG}_p64s_bench.json > $OUT/${TAG}_p64_pmc_summary.txt 2>&1
# 7. the conv path's chip-level leg as its own program: un-profiled wall figure, then the union of the same launches' execution
#    intervals from a kernel trace (the two must agree)      -> <tag>_conv_path_chip.txt
rm -rf /tmp/prof_cp
(echo "# un-profiled: python3 tools/conv_path_chip_trace.py --reps 200"; python3 $R/tools/conv_path_chip_trace.py --reps 200 2> /dev/null | tail -1
 echo "# rocprofv3 --kernel-trace -- python3 tools/conv_path_chip_trace.py --reps 200: the program's own wall figure under the profiler"
 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_cp -o r -- python3 $R/tools/conv_path_chip_trace.py --reps 200 2> /dev/null | tail -1
 echo "# ... and the union of the conv-path dispatches' execution intervals in that trace (tools/conv_path_union.py)"
 python3 $R/tools/conv_path_union.py $(find /tmp/prof_cp -name '*kernel_trace.csv' | head -1)) > $OUT/${TAG}_conv_path_chip.txt 2>&1 < /dev/null
cut -c1-200 $OUT/${TAG}_bench.json
head -14 $OUT/${TAG}_s1_kernel_stats.csv | cut -c1-150
tail -15 $OUT/${TAG}_s1_pmc_summary.txt
