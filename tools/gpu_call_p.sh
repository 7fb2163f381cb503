#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2p; mkdir -p $O
timeout 300 python tools/p8_check.py ringcheck > $O/ringcheck.log 2>&1; echo "ringcheck rc=$?"; grep -c " ok" $O/ringcheck.log; grep "FAIL" $O/ringcheck.log | head -20; tail -1 $O/ringcheck.log
for b in 32 64 128 256; do echo "== batch $b"; timeout 200 python tools/p8_check.py ring $b 2>&1 | grep -v amdgpu.ids | tee $O/ring_$b.log; done
