#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2f; mkdir -p $O
echo "== compaction test"; timeout 300 python -m pytest tests/test_engine_gpu.py -m gpu -q -k "compaction" > $O/t1.log 2>&1; echo "rc=$?"; tail -3 $O/t1.log
echo "== wgrad bench"; timeout 240 python tools/p8_check.py wgrad 256 > $O/wgrad_bench.log 2>&1; echo "rc=$?"; tail -8 $O/wgrad_bench.log
echo "== CU-sharing sweep (NT persistent WGs, TN persistent WGs)"
for cfg in "256 256" "192 64" "176 80" "160 96" "128 128" "208 48" "256 64"; do
  set -- $cfg
  VLB_GEMM_P8_WGS=$1 VLB_GEMM_TN8_WGS=$2 timeout 200 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-phase-times > $O/sweep_$1_$2.log 2>&1
  echo "p8_wgs=$1 tn8_wgs=$2 rc=$? $(tail -1 $O/sweep_$1_$2.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])' 2>/dev/null)"
done
echo "== same with wgrad stream off"; VLB_WGRAD_STREAM=0 timeout 200 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-phase-times > $O/nostream.log 2>&1; tail -1 $O/nostream.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])'
