#!/bin/bash
# Build tools/ab/libdbg.so = the working tree's csrc/ with --debug-variants (LD_CONV_DEBUG ablation / trace kernels and
# the shelved experiments), without touching the product library.  Run with LD_LIB_OVERRIDE=/root/repo/tools/ab/libdbg.so.
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
TMP=/tmp/ld_dbg_build
mkdir -p $TMP/localdiffusion-hallucination_amd/csrc $TMP/include $TMP/tools/experiments
cp $ROOT/localdiffusion-hallucination_amd/csrc/*.hip $ROOT/localdiffusion-hallucination_amd/csrc/*.h $ROOT/localdiffusion-hallucination_amd/csrc/build.sh $TMP/localdiffusion-hallucination_amd/csrc/
cp $ROOT/include/*.h $TMP/include/
cp $ROOT/tools/experiments/*.hip $TMP/tools/experiments/
(cd $TMP/localdiffusion-hallucination_amd/csrc && bash build.sh --debug-variants)
cp $TMP/localdiffusion-hallucination_amd/csrc/liblocaldiff_hip.so $ROOT/tools/ab/libdbg.so
echo "built $ROOT/tools/ab/libdbg.so"
