#!/bin/bash
# Runs ON THE GPU BOX: targeted tests, the default bench step A/B against a saved build, and env-switched variants of the tree build
set -u
V="${1:-ab_old}"; K="${2:-layernorm or attention}"
O=gpurun_out/ab2; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 600 python -m pytest tests -m gpu -x -q -k "$K" 2>&1 | tail -6 ) > $O/tests.log 2>&1; tail -7 $O/tests.log
one() { python bench.py --no-cpu-baseline --no-phase-times 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['roofline']['achieved'], d['roofline']['gemm_ms_per_step'], d['loss'])"; }
for i in 1 2; do
  unset VLB_LIB_PATH; one tree
  VLB_LN_BWD4=2 one tree_lnbwd_rif2
  export VLB_LIB_PATH=$PWD/vl-bert_amd/csrc/$V/libvlbert_hip.so; one $V
done 2>&1 | tee $O/ab.log
unset VLB_LIB_PATH
for b in 128 64 32; do python bench.py --no-cpu-baseline --global-batch $b --no-phase-times 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch $b', d['ms_per_step'], d['roofline']['frac'])"; done | tee $O/small.log
cd /tmp && export TMPDIR=/tmp
VLB_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/tr -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-phase-times > $GRAFT_REPO_ROOT/$O/tr.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/kstats.py $O/tr 5 16 | tee $O/kstats.txt; rm -rf $O/tr
