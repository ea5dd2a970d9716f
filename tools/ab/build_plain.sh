#!/bin/bash
# Build tools/ab/libplain.so = the working tree's csrc/ with --plain-stores (output stores left dirty in L2: the state before
# finding 98), without touching the product library.  Run with LD_LIB_OVERRIDE=$GRAFT_REPO_ROOT/tools/ab/libplain.so.
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
TMP=/tmp/ld_plain_build
mkdir -p $TMP/localdiffusion-hallucination_amd/csrc $TMP/include
cp $ROOT/localdiffusion-hallucination_amd/csrc/*.hip $ROOT/localdiffusion-hallucination_amd/csrc/*.h $ROOT/localdiffusion-hallucination_amd/csrc/build.sh $TMP/localdiffusion-hallucination_amd/csrc/
cp $ROOT/include/*.h $TMP/include/
(cd $TMP/localdiffusion-hallucination_amd/csrc && bash build.sh ${1:---plain-stores})
cp $TMP/localdiffusion-hallucination_amd/csrc/liblocaldiff_hip.so $ROOT/tools/ab/${2:-libplain.so}
echo "built $ROOT/tools/ab/${2:-libplain.so}"
