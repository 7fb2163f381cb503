#!/bin/bash
# uneven TN8 cut: parity, micro-bench, step A/B on one box
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_ops_gpu.py -q -x -k "wgrad" 2>&1 | tail -5
timeout 200 python tools/p8_check.py wgradgroup 256 128 64 32 2>&1 | tail -20
for b in 256 32; do for u in 0 1 0 1; do
  VLB_GEMM_TN8_UNEVEN=$u timeout 200 python bench.py --global-batch $b --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('batch $b uneven $u ms_per_step', j['ms_per_step'], 'value', j['value'])"
done; done
