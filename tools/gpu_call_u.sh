#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2u; mkdir -p $O
timeout 600 python tools/p8_check.py check > $O/check.log 2>&1; echo "check rc=$?"; grep -c " ok" $O/check.log; grep "FAIL" $O/check.log | head -20; tail -2 $O/check.log
for b in 128 32 256; do echo "== batch $b"; timeout 250 python tools/p8_check.py bench $b 2>&1 | grep -v amdgpu.ids | tee $O/shapes_$b.log; done
