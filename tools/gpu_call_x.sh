#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2x; mkdir -p $O
show='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["config"]["global_batch"], "ms", d["ms_per_step"], "samples/s", d["value"], "gemm TF", r["achieved"], "frac", r["frac"], "stepfrac", r["step_frac_of_peak"], "fwd", d["fwd_ms"], "fwd+bwd", d["fwd_bwd_ms"], "cpu", (d.get("cpu_baseline") or {}).get("value"))'
timeout 600 python bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | python -c "$show"
for gb in 128 64 32; do timeout 240 python bench.py --global-batch $gb --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$gb.log 2>&1; tail -1 $O/bench_$gb.log | python -c "$show"; done
timeout 400 python bench.py --large --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_large.log 2>&1; echo "large:"; tail -1 $O/bench_large.log | python -c "$show"
timeout 400 python bench.py --e2e --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_e2e.log 2>&1; echo "e2e 8+8:"; tail -1 $O/bench_e2e.log | python -c "$show"
