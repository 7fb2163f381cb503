#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2o; mkdir -p $O
for b in 128 64 32; do echo "== batch $b"; timeout 200 python tools/p8_check.py bench $b 2>&1 | grep -v amdgpu.ids | tee $O/shapes_$b.log; done
for gb in 64 32; do
  timeout 200 python bench.py --global-batch $gb --steps 20 --warmup 5 --no-cpu-baseline --no-phase-times > $O/bench_$gb.log 2>&1
  tail -1 $O/bench_$gb.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("gb", d["config"]["global_batch"], "ms", d["ms_per_step"], "samples/s", d["value"], "gemm ms", r["gemm_ms_per_step"], json.dumps(r["by_op"]))'
done
