#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2l; mkdir -p $O
timeout 200 python tools/p8_check.py ablate 256 2>&1 | grep -v amdgpu.ids | tee $O/ablate.log
timeout 200 python tools/p8_check.py bench 256 2>&1 | grep -v amdgpu.ids | tee $O/bench_shapes.log
