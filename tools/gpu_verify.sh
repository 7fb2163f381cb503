#!/bin/bash
# Runs ON THE GPU BOX: the round-end checks in one call -- full `-m gpu` suite, smoke(), default bench (JSON kept in gpurun_out/),
# plain-wrapper e2e bench, and `--gpus 2` (two ranks sharing the box's one GPU over gloo: a functional check of the self-launch + DP path).
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -2
timeout 300 python bench.py 2>/dev/null | tee $O/bench_default.json | cut -c1-260
timeout 200 python bench.py --e2e --aux-batch 0 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c1-260
timeout 200 python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | cut -c1-260
