#!/bin/bash
# Runs ON THE GPU BOX: kernel traces of the secondary bench modes (--large, --vcr, --vqa) -> gpurun_out/summary/<tag>_{large,vcr,vqa}_kernel_stats.txt
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
VLB_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats -d "$OUT/final_large_trace" -o r -- python $ROOT/bench.py --large --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-phase-times --no-clock-probe > "$OUT/final_large_trace.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$OUT/final_vcr_trace" -o r -- python $ROOT/bench.py --vcr --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/final_vcr_trace.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$OUT/final_vqa_trace" -o r -- python $ROOT/bench.py --vqa --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/final_vqa_trace.log" 2>&1
mkdir -p "$OUT/summary"
python "$ROOT/tools/profile_report.py" "$OUT" "$OUT/summary" "${1:-r04}" 5 extra-only
rm -rf "$OUT/final_large_trace" "$OUT/final_vcr_trace" "$OUT/final_vqa_trace"
grep '"metric"' "$OUT/final_large_trace.log" | tail -1 | cut -c1-220
grep '"metric"' "$OUT/final_vcr_trace.log" | tail -1 | cut -c1-220
grep '"metric"' "$OUT/final_vqa_trace.log" | tail -1 | cut -c1-220
