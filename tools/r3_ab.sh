#!/bin/bash
# Runs ON THE GPU BOX: targeted parity tests for a kernel change, then the default bench step A/B against a saved build
# (vl-bert_amd/csrc/<variant>/libvlbert_hip.so), then one kernel trace of the tree build.   tools/r3_ab.sh <variant dir> "<pytest -k expr>"
set -u
V="${1:-ab_old}"; K="${2:-layernorm or attention}"
O=gpurun_out/ab; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 600 python -m pytest tests -m gpu -x -q -k "$K" 2>&1 | tail -6 ) > $O/tests.log 2>&1; tail -7 $O/tests.log
bash tools/ab_bench.sh "$V" 2 2>&1 | tee $O/ab.log
cd /tmp && export TMPDIR=/tmp
VLB_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/tr -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-phase-times > $GRAFT_REPO_ROOT/$O/tr.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/kstats.py $O/tr 5 22 | tee $O/kstats.txt; rm -rf $O/tr
