#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): rocprofv3 passes over the default bench workload; summaries -> gpurun_out/summary/<tag>_*
# (copy them into profiles/), logs -> gpurun_out/final_*.log.  Usage: VLB_COMMIT=<hash> tools/make_profiles.sh [tag]   (the hash stamps <tag>_gemm_traffic.json)
#   1. --kernel-trace --stats            : per-kernel time
#   2. --kernel-trace --pmc <SQ set>     : wave / MFMA / LDS counters          (separate pass, kernel-trace only)
#   3. --kernel-trace --pmc FETCH_SIZE   : HBM-side read bytes                 (separate pass)
#   4. --kernel-trace --pmc WRITE_SIZE   : HBM-side write bytes                (separate pass)
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
# VLB_WGRAD_STREAM=0: the weight gradients normally run on a second stream and overlap the dgrad chain, which inflates every
# per-kernel duration in a trace; the kernel-efficiency profile (like the instrumented step inside bench.py that produces
# roofline.achieved) serialises them.  `value` in the default bench run is measured WITH the overlap.
export VLB_WGRAD_STREAM=0
CMD="python $ROOT/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-phase-times --no-clock-probe"
rocprofv3 --kernel-trace --stats -d "$OUT/final_trace" -o r -- $CMD > "$OUT/final_trace.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY -d "$OUT/final_sq" -o r -- $CMD > "$OUT/final_sq.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/final_fetch" -o r -- $CMD > "$OUT/final_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/final_write" -o r -- $CMD > "$OUT/final_write.log" 2>&1
# e2e configuration (C3: ResNet-101 trunk + ROIAlign + layer4 head in front of the same step); VLB_PROFILE_SKIP_E2E=1 skips these four passes
export VLB_VISION_WGRAD_STREAM=0
if [ "${VLB_PROFILE_SKIP_E2E:-0}" != 1 ]; then
rocprofv3 --kernel-trace --stats -d "$OUT/final_e2e_trace" -o r -- $CMD --e2e > "$OUT/final_e2e_trace.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY -d "$OUT/final_e2e_sq" -o r -- $CMD --e2e > "$OUT/final_e2e_sq.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/final_e2e_fetch" -o r -- $CMD --e2e > "$OUT/final_e2e_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/final_e2e_write" -o r -- $CMD --e2e > "$OUT/final_e2e_write.log" 2>&1
fi
# summarise on the box and drop the raw rocpd databases (five of them exceed the 64 MiB that gpurun copies back)
mkdir -p "$OUT/summary"
TAG="${1:-r05}"
python "$ROOT/tools/profile_report.py" "$OUT" "$OUT/summary" "$TAG"
rm -rf "$OUT"/final_trace "$OUT"/final_sq "$OUT"/final_fetch "$OUT"/final_write "$OUT"/final_e2e_trace "$OUT"/final_e2e_sq "$OUT"/final_e2e_fetch "$OUT"/final_e2e_write
grep '"metric"' "$OUT/final_trace.log" | tail -1 | cut -c1-200
grep '"metric"' "$OUT/final_e2e_trace.log" | tail -1 | cut -c1-200

# ---- optional sets ------------------------------------------------------------------------------------------------------------
one() {   # name, steps-in-process, note, command...
  local name=$1 steps=$2 note=$3; shift 3
  rocprofv3 --kernel-trace --stats -d "$OUT/tr_$name" -o r -- "$@" > "$OUT/tr_$name.log" 2>&1
  { echo "# $note"; echo "# $*"; grep '"metric"' "$OUT/tr_$name.log" | tail -1 | sed 's/^/# bench line of the traced run: /'; python "$ROOT/tools/kstats.py" "$OUT/tr_$name" "$steps" 45; } > "$OUT/summary/${TAG}_$name.txt"
  rm -rf "$OUT/tr_$name"
  head -6 "$OUT/summary/${TAG}_$name.txt" | cut -c1-160 | tail -3
}
for set in "${@:2}"; do
  case $set in
    small)
      for b in 64 32; do
        VLB_WGRAD_STREAM=0 one kernel_stats_batch$b 5 "per-GPU batch $b of the global-256 strong-scaling run on ONE MI355X (no communication), weight-gradient stream serialised; 1 warm-up + 3 timed + 1 instrumented step" \
          python "$ROOT/bench.py" --global-batch $b --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-phase-times --no-clock-probe
      done
      one vqa_fp32_kernel_stats 4 "BASELINE config 4 at its named precision: bench.py --vqa --precision fp32 (fp32 encoder, fp16 build around it); 1 warm-up + 2 timed + 1 instrumented optimizer step" \
        python "$ROOT/bench.py" --vqa --precision fp32 --steps 2 --warmup 1 --no-cpu-baseline
      VLB_WGRAD_STREAM=0 one f16_kernel_stats 5 "headline workload on the fp16 build of the library (VLB_PRECISION=f16): same kernels, IEEE fp16 as the 16-bit type" \
        python "$ROOT/bench.py" --precision f16 --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-phase-times --no-clock-probe ;;
    extra)
      VLB_WGRAD_STREAM=0 one large_kernel_stats 5 "bench.py --large (24 x 1024, S = 229, batch 64), weight-gradient stream serialised" \
        python "$ROOT/bench.py" --large --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-phase-times --no-clock-probe
      one vcr_kernel_stats 4 "bench.py --vcr (BASELINE config 5 through the module mirror)" python "$ROOT/bench.py" --vcr --steps 2 --warmup 1 --no-cpu-baseline
      one vqa_kernel_stats 4 "bench.py --vqa (config 4's workload, 16-bit build)" python "$ROOT/bench.py" --vqa --steps 2 --warmup 1 --no-cpu-baseline ;;
  esac
done
