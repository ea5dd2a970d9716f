#!/bin/bash
# Same-box A/B of WHOLE TREES (when the C ABI changed between the revisions, LD_LIB_OVERRIDE cannot swap libraries):
# export a git revision into tools/ab/old_tree/ (git-ignored, travels with gpurun), build its library in place, then on the
# GPU box alternate `python tools/ab/old_tree/bench.py` and `python bench.py` (tools/ab/tree_ab.sh).
#   usage: bash tools/ab/export_tree.sh <rev>
set -e
REV=${1:?revision}
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
DST=$ROOT/tools/ab/old_tree
rm -rf $DST && mkdir -p $DST
git -C $ROOT archive $REV bench.py localdiffusion_hallucination_amd.py localdiffusion-hallucination_amd include oracle profiles | tar -x -C $DST
bash $DST/localdiffusion-hallucination_amd/csrc/build.sh > /dev/null
echo "exported $REV to $DST"
