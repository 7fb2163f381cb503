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
python "$ROOT/tools/profile_report.py" "$OUT" "$OUT/summary" "${1:-r04}"
rm -rf "$OUT"/final_trace "$OUT"/final_sq "$OUT"/final_fetch "$OUT"/final_write "$OUT"/final_e2e_trace "$OUT"/final_e2e_sq "$OUT"/final_e2e_fetch "$OUT"/final_e2e_write
grep '"metric"' "$OUT/final_trace.log" | tail -1 | cut -c1-200
grep '"metric"' "$OUT/final_e2e_trace.log" | tail -1 | cut -c1-200
