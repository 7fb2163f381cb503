#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2v; mkdir -p $O
for m in 0 1 2; do VLB_LN_BWD4=$m timeout 120 python tools/ln_bench.py 25856 768 2>&1 | tail -1; done
for m in 0 1 2; do VLB_LN_BWD4=$m timeout 120 python tools/ln_bench.py 14656 1024 2>&1 | tail -1; done
for m in 0 1; do VLB_LN_BWD4=$m timeout 120 python tools/ln_bench.py 3232 768 2>&1 | tail -1; done
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "layernorm or ln" > $O/ln_tests.log 2>&1; echo "ln tests rc=$?"; tail -2 $O/ln_tests.log
