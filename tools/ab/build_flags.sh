#!/bin/bash
# Build tools/ab/<name>.so = the working tree's csrc/ with extra hipcc flags for some (or all) files, without touching the product
# library.  usage: bash tools/ab/build_flags.sh <name.so> "<extra flags>" [file1 file2 ... | all]
# e.g.  bash tools/ab/build_flags.sh libilp_conv.so "-mllvm -amdgpu-sched-strategy=max-ilp" conv3x3 conv3x3_s32
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
NAME=$1; EXTRA=$2; shift 2
FILES="${@:-all}"
TMP=/tmp/ld_flags_build_$$
mkdir -p $TMP/localdiffusion-hallucination_amd/csrc $TMP/include
cp $ROOT/localdiffusion-hallucination_amd/csrc/*.hip $ROOT/localdiffusion-hallucination_amd/csrc/*.h $ROOT/localdiffusion-hallucination_amd/csrc/build.sh $TMP/localdiffusion-hallucination_amd/csrc/
cp $ROOT/include/*.h $TMP/include/
cd $TMP/localdiffusion-hallucination_amd/csrc
if [ "$FILES" = all ]; then
  sed -i "s|^FLAGS=\"\(.*\)\"$|FLAGS=\"\1 $EXTRA\"|" build.sh
else
  for f in $FILES; do sed -i "s|    if \[ \$f = conv3x3 \]; then EXTRA=\"\"; fi|    if [ \$f = conv3x3 ]; then EXTRA=\"\"; fi\n    if [ \$f = $f ]; then EXTRA=\"\$EXTRA $EXTRA\"; fi|" build.sh; done
fi
bash build.sh > /dev/null
cp liblocaldiff_hip.so $ROOT/tools/ab/$NAME
rm -rf $TMP
echo "built $ROOT/tools/ab/$NAME ($FILES: $EXTRA)"
