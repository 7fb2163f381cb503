#!/bin/bash
# Builds libvlbert_hip.so for gfx950 (cross-compiles without a GPU).  Usage: build.sh [out_dir]
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${1:-$HERE/..}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=fast"
mkdir -p "$HERE/obj"
pids=()
for f in api gemm gemm_p8 gemm_tn8 layernorm embed loss attention optim roi_align vision; do
  [ -f "$HERE/$f.hip" ] || continue
  if [ ! -f "$HERE/obj/$f.o" ] || [ "$HERE/$f.hip" -nt "$HERE/obj/$f.o" ] || [ "$HERE/vlb_common.h" -nt "$HERE/obj/$f.o" ] || [ "$HERE/gemm_params.h" -nt "$HERE/obj/$f.o" ] \
     || [ "$HERE/../../include/vlbert_hip.h" -nt "$HERE/obj/$f.o" ]; then
    $HIPCC $FLAGS -c "$HERE/$f.hip" -o "$HERE/obj/$f.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/libvlbert_hip.so" "$HERE"/obj/*.o
echo "built $OUT/libvlbert_hip.so"
