#!/bin/bash
# Compiles the REFERENCE's own CPU ROIAlign (common/lib/roi_pooling/cpu/ROIAlign_cpu.cpp, unmodified, where it lies under
# /root/reference) + the C entry point of oracle/ref_shim/ into oracle/_ref/libroi_align_ref.so with g++ directly -- the
# reference's setup.py (torch cpp_extension + CUDA sources) is not used.  oracle/_ref/ is git-ignored but travels to the GPU box.
# Only runs where the reference tree exists; exits 0 with a note otherwise.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REF="${VLBERT_REFERENCE_ROOT:-/root/reference}"
SRC="$REF/common/lib/roi_pooling/cpu/ROIAlign_cpu.cpp"
OUT="$HERE/_ref"
if [ ! -f "$SRC" ]; then echo "oracle/build_ref.sh: $SRC not present -- nothing to build"; exit 0; fi
mkdir -p "$OUT"
if [ -f "$OUT/libroi_align_ref.so" ] && [ "$OUT/libroi_align_ref.so" -nt "$HERE/ref_shim/roi_align_ref_capi.cpp" ] && [ "$OUT/libroi_align_ref.so" -nt "$SRC" ]; then
  echo "oracle/_ref/libroi_align_ref.so is up to date"; exit 0
fi
read -r INC LIBDIR ABI <<< "$(python - <<'PY'
import sysconfig, torch
from torch.utils.cpp_extension import include_paths, library_paths
inc = " ".join("-I" + p for p in include_paths()) + " -I" + sysconfig.get_paths()["include"]
print(inc.replace(" ", "@"), library_paths()[0], int(torch._C._GLIBCXX_USE_CXX11_ABI))
PY
)"
INC="${INC//@/ }"
FLAGS="-O2 -std=c++17 -fPIC -w -D_GLIBCXX_USE_CXX11_ABI=$ABI -DTORCH_API_INCLUDE_EXTENSION_H $INC -I$REF/common/lib/roi_pooling"
g++ $FLAGS -include "$HERE/ref_shim/compat.h" -c "$SRC" -o "$OUT/ROIAlign_cpu.o" &
g++ $FLAGS -c "$HERE/ref_shim/roi_align_ref_capi.cpp" -o "$OUT/roi_align_ref_capi.o" &
wait
g++ -shared -o "$OUT/libroi_align_ref.so" "$OUT/ROIAlign_cpu.o" "$OUT/roi_align_ref_capi.o" -L"$LIBDIR" -ltorch -ltorch_cpu -lc10 -Wl,-rpath,"$LIBDIR"
rm -f "$OUT"/*.o
echo "built $OUT/libroi_align_ref.so"
