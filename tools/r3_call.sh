#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): tools/r3_call.sh <tag> <what...>   -- development calls of round 3; logs -> gpurun_out/<tag>/
set -u
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
for what in "$@"; do
  case $what in
    tests)   timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > $OUT/tests.log; tail -5 $OUT/tests.log ;;
    newtests) timeout 1500 python -m pytest tests -m gpu -q -k "24_layers or large_tile_kernels or batch_32 or ring or large_tile_core or layernorm_residual or sharded or overflow or two_ranks" 2>&1 | tail -60 > $OUT/newtests.log; tail -15 $OUT/newtests.log ;;
    dptests) timeout 900 python -m pytest tests/test_dp_gpu.py -m gpu -q 2>&1 | tail -30 > $OUT/dptests.log; tail -6 $OUT/dptests.log; grep -h "^rank" gpurun_out/dp_test_dp2_mirror*.log | tail -6; grep -h "Error\|assert" gpurun_out/dp_test_dp2_mirror*.log | head -8 ;;
    vqadp)   python bench.py --vqa --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/vqa_dp2.json 2> $OUT/vqa_dp2.err; cut -c1-400 $OUT/vqa_dp2.json; grep -v Gloo $OUT/vqa_dp2.err | tail -5 ;;
    f16)     timeout 1200 python -m pytest tests/test_f16_build_gpu.py -m gpu -q 2>&1 | tail -30 > $OUT/f16.log; tail -8 $OUT/f16.log; grep -E "Frobenius|passed|failed|FAILED|Error" gpurun_out/f16_suite.log | tail -30 ;;
    t24)     timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -q -s -k "24_layers" > $OUT/t24.log 2>&1; grep -E "Frobenius|passed|failed|per-layer|grad_norm|rel-fro|Error|assert" $OUT/t24.log | tail -25 ;;
    vcrtests) timeout 900 python -m pytest tests/test_vcr_gpu.py -m gpu -q -x -s > $OUT/vcrtests.log 2>&1; grep -E "passed|failed|FAILED|Error|fused clip|vcr " $OUT/vcrtests.log | tail -20 ;;
    vistests) timeout 900 python -m pytest tests/test_vision_gpu.py -m gpu -q -s > $OUT/vistests.log 2>&1; grep -E "passed|failed|FAILED|median|max rel-fro|grad-norm|gradient norm" $OUT/vistests.log | tail -30 ;;
    f32tests) timeout 1500 python -m pytest tests/test_f32_encoder_gpu.py -m gpu -q -x -s > $OUT/f32tests.log 2>&1; grep -E "passed|failed|FAILED|Error|error|max rel err|Frobenius|fp32 encoder|assert" $OUT/f32tests.log | tail -40 ;;
    vqa32)   python bench.py --vqa --precision fp32 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/vqa_fp32.json 2> $OUT/vqa_fp32.err; cut -c1-2500 $OUT/vqa_fp32.json; tail -3 $OUT/vqa_fp32.err
             python bench.py --vqa --precision f16 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/vqa_f16.json 2> $OUT/vqa_f16.err; cut -c1-300 $OUT/vqa_f16.json; tail -3 $OUT/vqa_f16.err ;;
    prof32)  cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof32 -o r -- python $GRAFT_REPO_ROOT/bench.py --vqa --precision fp32 --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/prof32.log 2>&1; cd $GRAFT_REPO_ROOT; python tools/kstats.py $OUT/prof32 4 30 | tee $OUT/prof32_kstats.txt; rm -rf $OUT/prof32 ;;
    bench)   python bench.py --no-cpu-baseline > $OUT/bench256.json 2> $OUT/bench256.err; cut -c1-400 $OUT/bench256.json ;;
    small)   for b in 128 64 32; do python bench.py --no-cpu-baseline --global-batch $b --no-phase-times > $OUT/bench$b.json 2> $OUT/bench$b.err; python -c "import json;d=json.load(open('$OUT/bench$b.json'));print($b, d['ms_per_step'], d['roofline']['frac'], d['roofline']['by_op']['host_launch_ms_whole_step'])"; done ;;
    graph)   for b in 256 32; do python bench.py --no-cpu-baseline --global-batch $b --no-phase-times --graph > $OUT/benchg$b.json 2> $OUT/benchg$b.err; python -c "import json;d=json.load(open('$OUT/benchg$b.json'));print('graph',$b, d['ms_per_step'])"; done ;;
    dp2)     python bench.py --gpus 2 --no-cpu-baseline --steps 10 > $OUT/dp2.json 2> $OUT/dp2.err; cut -c1-1200 $OUT/dp2.json; tail -3 $OUT/dp2.err
             python bench.py --gpus 2 --no-cpu-baseline --steps 10 --no-graph > $OUT/dp2_eager.json 2> $OUT/dp2_eager.err; cut -c1-300 $OUT/dp2_eager.json
             python bench.py --gpus 2 --no-cpu-baseline --steps 10 --dp-mode allreduce > $OUT/dp2_ar.json 2> $OUT/dp2_ar.err; cut -c1-300 $OUT/dp2_ar.json ;;
    nccl1)   python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --no-cpu-baseline --steps 5 > $OUT/nccl1.json 2> $OUT/nccl1.err; cut -c1-300 $OUT/nccl1.json; tail -2 $OUT/nccl1.err ;;
    *) echo "unknown $what" ;;
  esac
done
