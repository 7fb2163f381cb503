#!/bin/bash
# round-5 first GPU call: root-cause run of the DDP replica divergence + full suite
set -u
ROOT="${GRAFT_REPO_ROOT:-$PWD}"; cd $ROOT
OUT=$ROOT/gpurun_out/r5a; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=4
run_mirror() {  # $1 = tag, rest = args
  tag=$1; shift
  timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) tools/dp2_mirror_check.py "$@" > $OUT/mirror_$tag.log 2>&1
  rc=$?
  echo "mirror $tag rc=$rc | $(grep -c 'after 2 DDP steps: True' $OUT/mirror_$tag.log) ranks identical | $(grep 'address order' $OUT/mirror_$tag.log | sed 's/.*starts): //' | tr '\n' ' ')"
}
( time ( for i in 1 2 3; do run_mirror old$i --address-order; done; for i in $(seq 1 ${MIRROR_N:-12}); do run_mirror new$i; done ) ) 2>&1 | tee $OUT/mirror_loop.txt
( time timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -40 ) > $OUT/tests.log 2>&1; tail -30 $OUT/tests.log
