#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2h; mkdir -p $O
echo "== p8 bench (skew)"; timeout 300 python tools/p8_check.py bench 256 > $O/p8_bench.log 2>&1; echo "rc=$?"; cat $O/p8_bench.log
for sk in 0 8 16 28; do
VLB_GEMM_P8_SKEW=$sk timeout 240 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-phase-times > $O/bench_skew$sk.log 2>&1; tail -1 $O/bench_skew$sk.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("skew '$sk':", d["ms_per_step"], d["value"])'
done
