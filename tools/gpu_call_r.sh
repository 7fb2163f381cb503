#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2r; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
for gb in 256 128; do
timeout 240 python bench.py --global-batch $gb --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$gb.log 2>&1; echo "rc=$?"; tail -1 $O/bench_$gb.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["config"]["global_batch"], d["ms_per_step"], d["value"], "gemm TF", r["achieved"], "gemm ms", r["gemm_ms_per_step"], "fwd", d["fwd_ms"], "fwd+bwd", d["fwd_bwd_ms"])'
done
