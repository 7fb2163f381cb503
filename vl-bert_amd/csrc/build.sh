#!/bin/bash
# Builds libvlbert_hip.so (bf16 build) and libvlbert_hip_f16.so (the same sources with -DVLB_ACT_F16: IEEE fp16 as the 16-bit type,
# vlb_common.h) for gfx950 (cross-compiles without a GPU).  Usage: build.sh [out_dir]
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${1:-$HERE/..}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=fast"
SRCS="api gemm gemm_p8 gemm_tn8 layernorm embed loss attention optim roi_align vision f32_path comm"
JOBS="${VLB_BUILD_JOBS:-$(nproc)}"
mkdir -p "$HERE/obj" "$HERE/obj_f16"
todo=()
for variant in bf16 f16; do
  if [ $variant = f16 ]; then OBJ="$HERE/obj_f16"; else OBJ="$HERE/obj"; fi
  for f in $SRCS; do
    [ -f "$HERE/$f.hip" ] || continue
    o="$OBJ/$f.o"
    if [ ! -f "$o" ] || [ "$HERE/$f.hip" -nt "$o" ] || [ "$HERE/vlb_common.h" -nt "$o" ] || [ "$HERE/gemm_params.h" -nt "$o" ] \
       || [ "$HERE/../../include/vlbert_hip.h" -nt "$o" ]; then
      todo+=("$variant:$f")
    fi
  done
done
if [ ${#todo[@]} -gt 0 ]; then
  printf '%s\n' "${todo[@]}" | xargs -P "$JOBS" -I{} bash -c '
    v="${1%%:*}"; f="${1##*:}"
    if [ "$v" = f16 ]; then obj="$2/obj_f16"; def="-DVLB_ACT_F16"; else obj="$2/obj"; def=""; fi
    '"$HIPCC $FLAGS"' $def -c "$2/$f.hip" -o "$obj/$f.o"' _ {} "$HERE"
fi
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/libvlbert_hip.so" "$HERE"/obj/*.o
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/libvlbert_hip_f16.so" "$HERE"/obj_f16/*.o
echo "built $OUT/libvlbert_hip.so $OUT/libvlbert_hip_f16.so"
