#!/bin/bash
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_vision_gpu.py -q -x -k "roi_align" 2>&1 | tail -5
timeout 100 python tools/roi_bwd_bench.py synthetic 2>&1 | tail -4
timeout 100 python tools/roi_bwd_bench.py large 2>&1 | tail -4
for u in 0 1 0 1; do
  VLB_ROI_BWD_GATHER=$u timeout 200 python bench.py --e2e --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('e2e gather $u ms_per_step', j['ms_per_step'], 'value', j['value'])"
done
timeout 600 python -m pytest tests/test_vision_gpu.py tests/test_vcr_gpu.py -q -x 2>&1 | tail -5
