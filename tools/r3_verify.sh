#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the round-3 verification call -- full `-m gpu` suite, smoke(), the default bench line,
# then the headline rocprofv3 passes (kernel stats, SQ PMC, FETCH / WRITE) stamped with the commit.   tools/r3_verify.sh <commit> [tag]
set -u
C="${1:-unknown}"; TAG="${2:-r03}"
O=gpurun_out/verify; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 ) > $O/tests.log 2>&1; tail -8 $O/tests.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -3 | tee $O/smoke.log
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-700 $O/bench_default.json
VLB_COMMIT=$C VLB_PROFILE_SKIP_E2E=1 timeout 900 bash tools/make_profiles.sh $TAG
ls gpurun_out/summary
