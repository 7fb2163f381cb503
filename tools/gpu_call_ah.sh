#!/bin/bash
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_engine_gpu.py -q -x -k "fast_rcnn or vqa" 2>&1 | tail -3
timeout 200 python tools/vqa_debug.py 2>&1 | grep -v Warn | grep "^it\|non-finite\|grad norm" | head -12
timeout 400 python bench.py --vqa --steps 5 --warmup 2 2>gpurun_out/vqa.err | tee gpurun_out/bench_vqa.json | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('vqa ms', j['ms_per_step'], 'value', j['value'], 'loss', j['loss'], 'roofline', j['roofline']['achieved'], j['roofline']['frac'], 'cpu', j.get('cpu_baseline'))"
tail -3 gpurun_out/vqa.err
