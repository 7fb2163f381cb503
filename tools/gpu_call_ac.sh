#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_vision_gpu.py tests/test_vcr_gpu.py -q -x 2>&1 | tail -5
