#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2e; mkdir -p $O
echo "== TN tests"; timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "wgrad_tn" > $O/tn_tests.log 2>&1; RC=$?; echo "rc=$RC"; tail -8 $O/tn_tests.log
if [ $RC -ne 0 ]; then export VLB_GEMM_TN8=0; echo "TN8 FAILED -> disabled for the rest"; fi
echo "== wgrad bench"; timeout 240 python tools/p8_check.py wgrad 256 > $O/wgrad_bench.log 2>&1; echo "rc=$?"; cat $O/wgrad_bench.log
echo "== gpu tests"; timeout 600 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -6 $O/tests.log
echo "== bench"; timeout 240 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.log 2>&1; echo "rc=$?"; tail -1 $O/bench.log | cut -c1-300
echo "== profile"; mkdir -p gpurun_out/summary; (cd /tmp && VLB_WGRAD_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/final_trace -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-phase-times > $GRAFT_REPO_ROOT/gpurun_out/final_trace.log 2>&1); echo "rc=$?"
python tools/profile_report.py gpurun_out gpurun_out/summary r2e > $O/report.log 2>&1; rm -rf gpurun_out/final_trace; head -45 gpurun_out/summary/r2e_kernel_stats.txt | cut -c1-150
