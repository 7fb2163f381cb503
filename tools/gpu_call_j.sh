#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2j; mkdir -p $O
echo "== TN tests"; timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "wgrad" > $O/tn_tests.log 2>&1; echo "rc=$?"; tail -3 $O/tn_tests.log
echo "== wgrad bench"; timeout 240 python tools/p8_check.py wgrad 256 2>&1 | tail -8
echo "== bench"; timeout 240 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.log 2>&1; echo "rc=$?"; tail -1 $O/bench.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["ms_per_step"], d["value"], "gemm TF", r["achieved"], "gemm ms", r["gemm_ms_per_step"], "fwd", d["fwd_ms"], "fwd+bwd", d["fwd_bwd_ms"], "traffic", r["traffic"], r["algorithmic_GB_per_launch"])'
