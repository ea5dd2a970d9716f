#!/bin/bash
# Same-box A/B: build tools/ab/libold.so from the csrc/ sources of a git revision (default HEAD) next to the working
# tree's library; run with LD_LIB_OVERRIDE=/root/repo/tools/ab/libold.so.   usage: bash tools/ab/build_old.sh [rev]
set -e
REV=${1:-HEAD}
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
TMP=/tmp/ld_old_build
rm -rf $TMP && mkdir -p $TMP/pkg/csrc $TMP/include
git -C $ROOT archive $REV localdiffusion-hallucination_amd/csrc include | tar -x -C $TMP
SRC=$TMP/localdiffusion-hallucination_amd/csrc
pids=()
for f in runtime collective pack conv3x3 conv3x3_c32 conv1x1 conv_image gn_apply linattn linattn_fused attention time_embed pointwise; do
  EXTRA="-mllvm -amdgpu-mfma-vgpr-form"; [ $f = conv3x3 ] && EXTRA=""
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -ffp-contract=off $EXTRA -c $SRC/$f.hip -o $TMP/$f.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $TMP/*.o -ldl -o $ROOT/tools/ab/libold.so
echo "built $ROOT/tools/ab/libold.so from $REV"
