#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2w; mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "layernorm" > $O/ln.log 2>&1; echo "ln rc=$?"; tail -2 $O/ln.log
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -x -q > $O/engine.log 2>&1; echo "engine rc=$?"; tail -2 $O/engine.log
timeout 300 python tools/dp2_check.py > $O/dp2.log 2>&1; echo "dp2 rc=$?"; tail -3 $O/dp2.log
show='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["config"]["global_batch"], "ms", d["ms_per_step"], "samples/s", d["value"], "gemm TF", r["achieved"], "fwd", d["fwd_ms"], "fwd+bwd", d["fwd_bwd_ms"])'
for gb in 256 128 64 32; do timeout 240 python bench.py --global-batch $gb --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$gb.log 2>&1; tail -1 $O/bench_$gb.log | python -c "$show"; done
