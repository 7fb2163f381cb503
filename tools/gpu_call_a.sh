#!/bin/bash
# round-2 GPU call A: new GEMM core check + bench, GPU test-suite, step bench, 2-rank launch on one device
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2a; mkdir -p $O
echo "== p8 check"; timeout 240 python tools/p8_check.py check > $O/p8_check.log 2>&1; RC=$?; echo "p8 check rc=$RC"; tail -5 $O/p8_check.log
if [ $RC -eq 0 ]; then
  echo "== p8 bench"; timeout 240 python tools/p8_check.py bench 256 > $O/p8_bench.log 2>&1; echo "rc=$?"; cat $O/p8_bench.log
  P8=1
else
  echo "p8 check FAILED/hung -> rest runs on the 128x128 kernels"; P8=0
  grep -c FAIL $O/p8_check.log; grep FAIL $O/p8_check.log | head -20
fi
export VLB_GEMM_P8=$P8
echo "== gpu tests (VLB_GEMM_P8=$P8)"; timeout 600 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -15 $O/tests.log
echo "== bench"; timeout 240 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.log 2>&1; echo "rc=$?"; tail -2 $O/bench.log
if [ $P8 -eq 1 ]; then
  VLB_GEMM_P8=0 timeout 240 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_p8off.log 2>&1; echo "rc=$?"; tail -1 $O/bench_p8off.log
fi
echo "== bench --gpus 2 on one device (gloo)"; timeout 300 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_g2.log 2>&1; echo "rc=$?"; tail -3 $O/bench_g2.log
echo "== dp2 check"; timeout 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tools/dp2_check.py > $O/dp2.log 2>&1; echo "rc=$?"; tail -6 $O/dp2.log
VLB_DP_WIRE=fp32 timeout 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 tools/dp2_check.py > $O/dp2_fp32.log 2>&1; echo "rc=$?"; tail -4 $O/dp2_fp32.log
