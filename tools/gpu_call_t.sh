#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2t; mkdir -p $O
timeout 300 python tools/p8_check.py p4check > $O/p4check.log 2>&1; echo "p4check rc=$?"; grep -c " ok" $O/p4check.log; grep "FAIL" $O/p4check.log | head -20; tail -2 $O/p4check.log
for b in 256 128; do echo "== batch $b"; timeout 250 python tools/p8_check.py p4 $b 2>&1 | grep -v amdgpu.ids | tee $O/p4_$b.log; done
