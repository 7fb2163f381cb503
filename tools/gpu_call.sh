#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): ONE script for every development / verification call.
#   tools/gpu_call.sh <tag> <what...>     logs -> gpurun_out/<tag>/ ; every step runs under its own `timeout`
# what:
#   tests            full `-m gpu` suite                        | tests:<-k expr>   a subset (quote the expression)
#   libtests:<dir>:<-k expr>   the same subset against a saved build vl-bert_amd/csrc/<dir>/libvlbert_hip.so
#   smoke            __graft_entry__.smoke()
#   bench            default bench line (with other_configs + the SURVEY-form CPU baseline)
#   ab:<dir>[,...]   default bench step, tree build interleaved with saved builds vl-bert_amd/csrc/<dir>/libvlbert_hip.so and
#                    env-switched variants given as <NAME=VALUE> items (e.g. ab:VLB_GEMM_P8_DRAIN=1,ab_d3)
#   gemm             tools/p8_check.py bench 256 (per-shape GEMM table, tile heights + model)
#   small            per-GPU batches 128 / 64 / 32 (strong-scaling columns, no communication)
#   modes            every secondary bench mode quoted in DESIGN.md §6 (f16, e2e, large, vqa, vqa fp32, vcr f16)
#   phase            tools/p8_phase_probe.py (in-kernel main-loop / epilogue split of the layer's NT GEMMs)
#   stream           tools/ln_bench.py: LayerNorm forward / backward variants, squared norm, AdamW, wire cast (the HBM-bound kernels of the step)
#   ntprobe          tools/nt_coherence_probe.hip (are non-temporal loads coherent with what other agents wrote since the last NT load?)
#   hbm              tools/hbm_stream_probe.hip (attainable streaming rates by access pattern: copy / read / fill / the AdamW stream mix)
#   tn8              tools/tn8_probe.py (weight-gradient core: matrix-instruction forms, in-kernel clock, ablations)  | embed  tools/embed_bench.py
#   e2e              config 3 eager / --graph / one vs three weight-gradient side streams     | clock   tools/clock_probe.py
#   dp2              bench.py --gpus 2 on this box (2 ranks sharing the GPU over gloo; over RCCL where the box has 2 GPUs)
#   trace            rocprofv3 --kernel-trace --stats of the headline workload -> kernel table (tools/kstats.py)
#   profiles[:tag]   tools/make_profiles.sh (kernel stats, SQ PMC, FETCH / WRITE passes; summaries in gpurun_out/summary)
#   mirror[:n]       tools/dp2_mirror_check.py n times (default 12) + 3 times with the pre-round-5 address-ordered norm (--address-order):
#                    the replica-divergence root cause of round 4's red GPU test
set -u
TAG=$1; shift
ROOT="${GRAFT_REPO_ROOT:-$PWD}"
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $ROOT
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline',{}); print('%-28s %8.3f ms %10.1f samples/s  gemm %7.1f TF frac %.4f gemm_ms %s loss %s' % ('$1', d['ms_per_step'], d['value'], r.get('achieved',0), r.get('frac',0), r.get('gemm_ms_per_step'), d.get('loss')))" 2>/dev/null || echo "$1 FAILED"; }
for what in "$@"; do
  case $what in
    tests)    ( time timeout 1700 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $OUT/tests.log 2>&1; tail -9 $OUT/tests.log ;;
    tests:*)  ( time timeout 1500 python -m pytest tests -m gpu -q -k "${what#tests:}" 2>&1 | tail -40 ) > $OUT/tests_subset.log 2>&1; tail -25 $OUT/tests_subset.log ;;
    libtests:*) spec="${what#libtests:}"; d="${spec%%:*}"; k="${spec#*:}"      # libtests:<variant dir>:<-k expr>: parity tests against a saved build
              ( time VLB_LIB_PATH=$ROOT/vl-bert_amd/csrc/$d/libvlbert_hip.so timeout 900 python -m pytest tests -m gpu -q -x -k "$k" 2>&1 | tail -15 ) > $OUT/libtests_$d.log 2>&1; tail -8 $OUT/libtests_$d.log ;;
    envtests:*) spec="${what#envtests:}"; e="${spec%%:*}"; k="${spec#*:}"       # envtests:<NAME=VALUE[,NAME=VALUE...]>:<-k expr>: a subset under environment switches
              ( time env ${e//,/ } timeout 900 python -m pytest tests -m gpu -q -x -k "$k" 2>&1 | tail -12 ) > $OUT/envtests.log 2>&1; tail -6 $OUT/envtests.log ;;
    smoke)    timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -3 | tee $OUT/smoke.log ;;
    bench)    timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cut -c1-900 $OUT/bench_default.json; tail -2 $OUT/bench_default.err
              python -c "import json; d=json.load(open('$OUT/bench_default.json')); print('cpu_baseline', d.get('cpu_baseline')); print('other_configs', json.dumps(d.get('other_configs'))[:1500])" ;;
    ab:*)     IFS=, read -ra V <<< "${what#ab:}"
              for i in 1 2; do
                unset VLB_LIB_PATH; timeout 300 python bench.py --no-cpu-baseline --no-phase-times --no-clock-probe 2>/dev/null | line tree
                for v in "${V[@]}"; do
                  if [[ $v == *=* ]]; then env "$v" timeout 300 python bench.py --no-cpu-baseline --no-phase-times --no-clock-probe 2>/dev/null | line "$v"
                  else VLB_LIB_PATH=$ROOT/vl-bert_amd/csrc/$v/libvlbert_hip.so timeout 300 python bench.py --no-cpu-baseline --no-phase-times --no-clock-probe 2>/dev/null | line "$v"; fi
                done
              done 2>&1 | tee $OUT/ab.log ;;
    gemm)     timeout 600 python tools/p8_check.py bench 256 2>&1 | tee $OUT/gemm_table.txt | tail -16 ;;
    gemm:*)   IFS=, read -ra V <<< "${what#gemm:}"
              for v in "${V[@]}"; do echo "== $v"; if [[ $v == *=* ]]; then env "$v" timeout 600 python tools/p8_check.py model 256; else VLB_LIB_PATH=$ROOT/vl-bert_amd/csrc/$v/libvlbert_hip.so timeout 600 python tools/p8_check.py model 256; fi; done 2>&1 | tee $OUT/gemm_variants.txt | tail -60 ;;
    small)    for b in 128 64 32; do timeout 300 python bench.py --no-cpu-baseline --global-batch $b --no-phase-times 2>/dev/null | tee $OUT/bench$b.json | line "batch $b"; done | tee $OUT/small.log ;;
    modes)    run() { n=$1; shift; timeout 600 python bench.py "$@" > $OUT/$n.json 2> $OUT/$n.err; line $n < $OUT/$n.json; }
              run f16 --precision f16 --no-cpu-baseline; run e2e --e2e --no-cpu-baseline; run large --large --no-cpu-baseline
              run large_f16 --large --precision f16 --no-cpu-baseline; run vqa --vqa --steps 5 --warmup 2 --no-cpu-baseline
              run vqa_fp32 --vqa --precision fp32 --steps 3 --warmup 1 --no-cpu-baseline; run vcr_f16 --vcr --precision f16 --steps 3 --warmup 1 --no-cpu-baseline
              run vcr --vcr --steps 3 --warmup 1 --no-cpu-baseline ;;
    e2e)      for v in "" "--graph" ; do timeout 400 python bench.py --e2e --no-cpu-baseline --no-phase-times --no-clock-probe $v 2>/dev/null | line "e2e $v"; done | tee $OUT/e2e.log
              VLB_VISION_WGRAD_STREAMS=1 timeout 400 python bench.py --e2e --no-cpu-baseline --no-phase-times --no-clock-probe 2>/dev/null | line "e2e 1 side stream" | tee -a $OUT/e2e.log
              VLB_VISION_WGRAD_STREAMS=1 timeout 400 python bench.py --e2e --no-cpu-baseline --no-phase-times --no-clock-probe --graph 2>/dev/null | line "e2e 1 side stream --graph" | tee -a $OUT/e2e.log ;;
    phase)    { timeout 200 python tools/p8_phase_probe.py 256 --spread; timeout 200 python tools/p8_phase_probe.py 32; } 2>&1 | grep -v amdgpu.ids | tee $OUT/p8_phase_probe.txt ;;
    stream)   for v in "VLB_LN_BWD4=1 VLB_LN_FWD_ROWS=1" "VLB_LN_BWD4=2 VLB_LN_FWD_ROWS=2" "VLB_LN_BWD4=1 VLB_LN_FWD_ROWS=4"; do env $v timeout 200 python tools/ln_bench.py 25856 768 $( [ "$v" = "VLB_LN_BWD4=1 VLB_LN_FWD_ROWS=1" ] || echo noopt ) 2>&1 | grep -v amdgpu.ids; done | tee $OUT/stream_bench.txt ;;
    ntprobe)  { hipcc --offload-arch=gfx950 -O3 tools/nt_coherence_probe.hip -o /tmp/nt_probe && timeout 300 /tmp/nt_probe; } 2>&1 | tee $OUT/nt_coherence_probe.txt ;;
    hbm)      { hipcc --offload-arch=gfx950 -O3 tools/hbm_stream_probe.hip -o /tmp/hbm_probe && timeout 200 /tmp/hbm_probe; } 2>&1 | tee $OUT/hbm_stream_probe.txt | awk '/GB\/s/ { if ($(NF-1) > best[$1]) { best[$1] = $(NF-1); line[$1] = $0 } } END { for (k in line) print "best", line[k] }' ;;
    tn8)      # round 6: the grouped weight-gradient launch, 16x16x32 vs 32x32x16 matrix instructions, clock stamps, ablations (needs csrc/ab built with -DVLB_TN8_PROBE)
              VLB_LIB_PATH=$ROOT/vl-bert_amd/csrc/ab/libvlbert_hip.so timeout 400 python tools/tn8_probe.py 256 --ablate 2>&1 | grep -v amdgpu.ids | tee $OUT/tn8_probe.txt ;;
    embed)    timeout 200 python tools/embed_bench.py 2>&1 | grep -v amdgpu.ids | tee $OUT/embed_bench.txt ;;
    clock)    timeout 300 python tools/clock_probe.py 4 2>&1 | grep -v amdgpu.ids | tee $OUT/clock_probe.txt ;;
    graphsmall) for b in 64 32; do timeout 300 python bench.py --no-cpu-baseline --global-batch $b --no-phase-times --graph 2>/dev/null | line "batch $b --graph"; done | tee $OUT/graphsmall.log ;;
    dp2)      timeout 600 python bench.py --gpus 2 --no-cpu-baseline --steps 10 > $OUT/dp2.json 2> $OUT/dp2.err; grep '^{' $OUT/dp2.json | cut -c1-1200; grep -v Gloo $OUT/dp2.err | tail -3
              grep -o '"comm": {[^}]*}' $OUT/dp2.json ;;
    trace)    ( cd /tmp && export TMPDIR=/tmp && VLB_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/tr -o r -- python $ROOT/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-phase-times --no-clock-probe > $OUT/tr.log 2>&1 )
              python tools/kstats.py $OUT/tr 5 24 | tee $OUT/kstats.txt; rm -rf $OUT/tr ;;
    profiles*) t="${what#profiles}"; t="${t#:}"; VLB_COMMIT=$(cat $ROOT/.commit_stamp 2>/dev/null || echo unknown) timeout 2000 bash tools/make_profiles.sh "${t:-r05}" small; ls $ROOT/gpurun_out/summary ;;
    mirror*)  n="${what#mirror}"; n="${n#:}"; n="${n:-12}"
              run_mirror() { tag=$1; shift
                timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) tools/dp2_mirror_check.py "$@" > $OUT/mirror_$tag.log 2>&1
                echo "mirror $tag rc=$? | $(grep -c 'after 2 DDP steps: True' $OUT/mirror_$tag.log) ranks identical | $(grep 'address order' $OUT/mirror_$tag.log | sed 's/.*starts): //' | tr '\n' ' ')"; }
              { for i in 1 2 3; do run_mirror old$i --address-order; done; for i in $(seq 1 $n); do run_mirror new$i; done; } 2>&1 | tee $OUT/mirror_loop.txt ;;
    *) echo "unknown step: $what" ;;
  esac
done
