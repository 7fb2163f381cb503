#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2m; mkdir -p $O
timeout 200 python tools/p8_check.py stagger 256 2>&1 | grep -v amdgpu.ids | tee $O/stagger.log
