#!/bin/bash
# A/B build: the whole library with extra compiler flags into vl-bert_amd/csrc/ab/libvlbert_hip.so (select with VLB_LIB_PATH).
# Usage: tools/build_variant.sh -DVLB_HASH_MUL32
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
SRC="$ROOT/vl-bert_amd/csrc"
OUT="$SRC/${VLB_AB_DIR:-ab}"
mkdir -p "$OUT/obj"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=fast $*"
pids=()
for f in api gemm gemm_p8 gemm_tn8 layernorm embed loss attention optim roi_align vision f32_path comm; do
  /opt/rocm/bin/hipcc $FLAGS -I"$ROOT/include" -c "$SRC/$f.hip" -o "$OUT/obj/$f.o" 2>/dev/null &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libvlbert_hip.so" "$OUT"/obj/*.o
echo "built $OUT/libvlbert_hip.so ($*)"
