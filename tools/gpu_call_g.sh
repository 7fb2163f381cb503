#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2g; mkdir -p $O
echo "== TN tests"; timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "wgrad" > $O/tn_tests.log 2>&1; RC=$?; echo "rc=$RC"; tail -5 $O/tn_tests.log
echo "== gpu tests"; timeout 600 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -6 $O/tests.log
echo "== bench"; timeout 240 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.log 2>&1; echo "rc=$?"; tail -1 $O/bench.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["ms_per_step"], d["value"], "gemm TF", r["achieved"], "gemm ms", r["gemm_ms_per_step"], "fwd", d["fwd_ms"], "fwd+bwd", d["fwd_bwd_ms"])'
VLB_WGRAD_STREAM=0 timeout 240 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-phase-times > $O/bench_nostream.log 2>&1; tail -1 $O/bench_nostream.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("no side stream:", d["ms_per_step"], d["value"])'
echo "== profile"; mkdir -p gpurun_out/summary; (cd /tmp && VLB_WGRAD_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/final_trace -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-phase-times > $GRAFT_REPO_ROOT/gpurun_out/final_trace.log 2>&1); echo "rc=$?"
python tools/profile_report.py gpurun_out gpurun_out/summary r2g > $O/report.log 2>&1; rm -rf gpurun_out/final_trace; head -40 gpurun_out/summary/r2g_kernel_stats.txt | cut -c1-150
