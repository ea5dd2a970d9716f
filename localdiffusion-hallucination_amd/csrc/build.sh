#!/bin/bash
# Build liblocaldiff_hip.so for gfx950 (cross-compiles without a GPU).
#   ./build.sh                   incremental: recompile the sources that are newer than their objects
#   ./build.sh --clean           remove build/ and the library first (what __graft_entry__.build() does by default)
#   ./build.sh --plain-stores    the A/B build of finding 98: plain instead of write-through (sc1) output stores
#   ./build.sh --debug-variants  also instantiate the LD_CONV_DEBUG ablation / trace kernels (-DLD_DEBUG_VARIANTS);
#                                the flag is recorded in build/.flags, so switching it rebuilds everything
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
# -amdgpu-kernarg-preload-count: leading scalar kernel arguments (up to 14 dwords) arrive in SGPRs with the wave -- no scalar
# round trip in front of the first requests (docs/findings.md 83); by-value structs are never preloaded
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=16"
CLEAN=0
for arg in "$@"; do
  case "$arg" in
    --clean) CLEAN=1 ;;
    --debug-variants) FLAGS="$FLAGS -DLD_DEBUG_VARIANTS" ;;
    --plain-stores) FLAGS="$FLAGS -DLD_STORE_WT=0" ;;
    --tile-xcd) FLAGS="$FLAGS -DLD_TILE_XCD=1" ;;            # experiment: rectangular tile regions per XCD in the 3x3 convolutions
    --load-nt) FLAGS="$FLAGS -DLD_LOAD_NT=1" ;;              # experiment: non-temporal activation loads in the 3x3 convolutions
    --wt-level=*) FLAGS="$FLAGS -DLD_STORE_WT=${arg#--wt-level=}" ;;   # experiments: 0 plain, 1 the 16-byte activation stores, 2 + the narrow ones      # A/B build: plain instead of write-through output stores (finding 98)
    *) echo "build.sh: unknown option $arg" >&2; exit 2 ;;
  esac
done
if [ $CLEAN = 1 ]; then rm -rf build liblocaldiff_hip.so; fi
mkdir -p build
if [ "$(cat build/.flags 2>/dev/null)" != "$FLAGS" ]; then rm -f build/*.o; echo "$FLAGS" > build/.flags; fi
pids=()
for f in runtime collective pack conv3x3 conv3x3_c32 conv3x3_s32 conv1x1 conv_image gn_apply linattn linattn_fused attention time_embed pointwise; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ common.hip.h -nt build/$f.o ] || [ ../../include/localdiff_hip.h -nt build/$f.o ] || [ -n "$(find . -maxdepth 1 -name '*.hip.h' -newer build/$f.o)" ]; then
    # MFMA results in VGPRs (no v_accvgpr_read/mov traffic in the epilogues that post-process accumulators);
    # the register-staged generic conv measured 1.5 % slower with it and keeps the default AGPR form
    EXTRA="-mllvm -amdgpu-mfma-vgpr-form"
    if [ $f = conv3x3 ]; then EXTRA=""; fi
    $HIPCC $FLAGS $EXTRA -c $f.hip -o build/$f.o &
    pids+=($!)
  fi
done
case "$FLAGS" in *LD_DEBUG_VARIANTS*)   # shelved experiments that a --debug-variants library can switch on
  for x in conv3x3_ksplit conv3x3_ring; do
    if [ ! -f build/$x.o ] || [ ../../tools/experiments/$x.hip -nt build/$x.o ] || [ common.hip.h -nt build/$x.o ]; then
      $HIPCC $FLAGS -mllvm -amdgpu-mfma-vgpr-form -c ../../tools/experiments/$x.hip -o build/$x.o &
      pids+=($!)
    fi
  done ;;
  *) rm -f build/conv3x3_ksplit.o build/conv3x3_ring.o ;;
esac
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC build/*.o -ldl -o liblocaldiff_hip.so
echo "built $(pwd)/liblocaldiff_hip.so"
