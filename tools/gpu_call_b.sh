#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2d; mkdir -p $O
echo "== p8 check"; timeout 240 python tools/p8_check.py check > $O/p8_check.log 2>&1; RC=$?; echo "p8 check rc=$RC"; grep -v " ok$" $O/p8_check.log | tail -12
[ $RC -eq 0 ] || exit 0
echo "== p8 bench"; timeout 300 python tools/p8_check.py bench 256 > $O/p8_bench.log 2>&1; echo "rc=$?"; cat $O/p8_bench.log
echo "== gpu tests"; timeout 600 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -12 $O/tests.log
echo "== bench"; timeout 240 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.log 2>&1; echo "rc=$?"; tail -1 $O/bench.log | cut -c1-400
