#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2i; mkdir -p $O
echo "== p8 check"; timeout 240 python tools/p8_check.py check > $O/p8_check.log 2>&1; RC=$?; echo "p8 check rc=$RC"; grep -v " ok$" $O/p8_check.log | tail -12
[ $RC -eq 0 ] || exit 0
echo "== ablate"; timeout 200 python tools/p8_check.py ablate 256 2>&1 | tail -8
echo "== p8 bench"; timeout 300 python tools/p8_check.py bench 256 > $O/p8_bench.log 2>&1; echo "rc=$?"; cut -c1-120 $O/p8_bench.log
echo "== gpu tests"; timeout 600 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
echo "== bench"; timeout 240 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.log 2>&1; echo "rc=$?"; tail -1 $O/bench.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["ms_per_step"], d["value"], "fwd", d["fwd_ms"], "fwd+bwd", d["fwd_bwd_ms"])'
