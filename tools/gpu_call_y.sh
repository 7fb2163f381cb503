#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2y; mkdir -p $O
timeout 900 python tools/p8_check.py check > $O/check.log 2>&1; echo "check rc=$?"; grep -c " ok" $O/check.log; grep "FAIL" $O/check.log | head -10; tail -1 $O/check.log
timeout 600 python -m pytest tests/test_vision_gpu.py -m gpu -x -q > $O/vision.log 2>&1; echo "vision rc=$?"; tail -2 $O/vision.log
show='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["config"]["global_batch"], "ms", d["ms_per_step"], "samples/s", d["value"], "gemm TF", r["achieved"], "frac", r["frac"])'
for i in 1 2; do
VLB_GEMM_P8=1 timeout 400 python bench.py --e2e --steps 10 --warmup 3 --no-cpu-baseline --no-phase-times 2>&1 | tail -1 | python -c "$show"
done
