#!/bin/bash
# Same-box A/B: build tools/ab/libold.so from the csrc/ sources of a git revision (default HEAD) with THAT revision's own
# build.sh (file list and flags), next to the working tree's library; run with LD_LIB_OVERRIDE=$GRAFT_REPO_ROOT/tools/ab/libold.so
# (tools/ab/lib_ab.sh libold.so).   usage: bash tools/ab/build_old.sh [rev] [build.sh options]
set -e
REV=${1:-HEAD}
shift || true
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
TMP=/tmp/ld_old_build
rm -rf $TMP && mkdir -p $TMP
git -C $ROOT archive $REV localdiffusion-hallucination_amd/csrc include | tar -x -C $TMP
(cd $TMP/localdiffusion-hallucination_amd/csrc && bash build.sh "$@")
cp $TMP/localdiffusion-hallucination_amd/csrc/liblocaldiff_hip.so $ROOT/tools/ab/libold.so
echo "built $ROOT/tools/ab/libold.so from $REV"
