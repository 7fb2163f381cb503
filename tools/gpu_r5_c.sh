#!/bin/bash
# round-5 GPU call: stream-K correctness + speed (quick A/B loop)
set -u
ROOT="${GRAFT_REPO_ROOT:-$PWD}"; cd $ROOT
OUT=$ROOT/gpurun_out/${1:-r5c}; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=4
( time timeout 600 python -m pytest tests -m gpu -q -x -k "stream_k or (layernorm_residual_fp16_stream and stream)" 2>&1 | tail -25 ) > $OUT/tests_sk.log 2>&1; tail -6 $OUT/tests_sk.log
timeout 300 python tools/sk_bench.py 32 64 128 2>&1 | grep -v amdgpu.ids | tee $OUT/sk_bench.txt
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline',{}); print('%-28s %8.3f ms %10.1f samples/s  gemm %7.1f TF frac %.4f gemm_ms %s loss %s' % ('$1', d['ms_per_step'], d['value'], r.get('achieved',0), r.get('frac',0), r.get('gemm_ms_per_step'), d.get('loss')))" 2>/dev/null || echo "$1 FAILED"; }
for b in ${SK_BATCHES:-32 64}; do for sk in 0 1; do VLB_GEMM_SK=$sk timeout 300 python bench.py --no-cpu-baseline --global-batch $b --no-phase-times --no-clock-probe 2>/dev/null | line "batch $b sk=$sk"; done; done | tee $OUT/small_sk.txt
