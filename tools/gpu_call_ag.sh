#!/bin/bash
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_vcr_gpu.py -q -x 2>&1 | tail -4
timeout 400 python bench.py --vqa --steps 3 --warmup 1 2>gpurun_out/vqa.err | tee gpurun_out/bench_vqa.json | cut -c1-1500
tail -5 gpurun_out/vqa.err
