#!/bin/bash
# GPU box: VERDICT r2 item 2(ii) -- the persistent C=32 convolution for the 4-patch sub-batches on HALF the chip, each
# sub-batch stream confined to four whole XCDs by a CU mask (same-box alternating runs of the bench step).
cd $GRAFT_REPO_ROOT
bash tools/ab/ab_env.sh "LD_X=0" "LD_BENCH_CU_MASK=xcd" "LD_BENCH_CU_MASK=xcd LD_CONV_C32_MIN_TILES=1024 LD_CONV_C32_CUS=128" "LD_CONV_C32_MIN_TILES=1024 LD_CONV_C32_CUS=128" "LD_BENCH_CU_MASK=xcd LD_CONV_C32_MIN_TILES=1024 LD_CONV_C32_CUS=256"
