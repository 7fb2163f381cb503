#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2q; mkdir -p $O
timeout 900 python -m pytest tests/test_vcr_gpu.py -m gpu -x -q -s > $O/vcr.log 2>&1; echo "vcr rc=$?"; grep -E "passed|failed|error" $O/vcr.log | tail -3; grep -E "^  vcr|^vcr|FusedSGD|sgd n|Error|assert" $O/vcr.log | head -60
timeout 900 python -m pytest tests/test_vision_gpu.py -m gpu -x -q -s -k "multitask or headline or shipped or e2e_step" > $O/vision.log 2>&1; echo "vision rc=$?"; grep -E "passed|failed|error" $O/vision.log | tail -3; grep -E "multitask e2e|ResNet-101|e2e engine|rel-fro|Error|assert|step " $O/vision.log | head -40
