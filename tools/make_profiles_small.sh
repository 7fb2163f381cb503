#!/bin/bash
# Runs ON THE GPU BOX: kernel traces of the per-GPU batches of a strong-scaling run (64 / 32 samples: the 4- / 8-GPU columns), of the fp32
# VQA step (BASELINE config 4 at its named precision) and of the fp16 build on the headline workload
#   -> gpurun_out/summary/<tag>_kernel_stats_batch{64,32}.txt, <tag>_vqa_fp32_kernel_stats.txt, <tag>_f16_kernel_stats.txt
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$ROOT/gpurun_out"
TAG="${1:-r04}"
cd /tmp && export TMPDIR=/tmp
mkdir -p "$OUT/summary"
one() {   # name, steps-in-process, note, command...
  local name=$1 steps=$2 note=$3; shift 3
  rocprofv3 --kernel-trace --stats -d "$OUT/tr_$name" -o r -- "$@" > "$OUT/tr_$name.log" 2>&1
  { echo "# $note"; echo "# $*"; grep '"metric"' "$OUT/tr_$name.log" | tail -1 | sed 's/^/# bench line of the traced run: /'; python "$ROOT/tools/kstats.py" "$OUT/tr_$name" "$steps" 45; } > "$OUT/summary/${TAG}_$name.txt"
  rm -rf "$OUT/tr_$name"
  head -6 "$OUT/summary/${TAG}_$name.txt" | cut -c1-160 | tail -3
}
for b in 64 32; do
  VLB_WGRAD_STREAM=0 one kernel_stats_batch$b 5 "per-GPU batch $b of the global-256 strong-scaling run on ONE MI355X (no communication), weight-gradient stream serialised; 1 warm-up + 3 timed + 1 instrumented step" \
    python "$ROOT/bench.py" --global-batch $b --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-phase-times --no-clock-probe
done
one vqa_fp32_kernel_stats 4 "BASELINE config 4 at its named precision: bench.py --vqa --precision fp32 (fp32 encoder, fp16 build around it); 1 warm-up + 2 timed + 1 instrumented optimizer step" \
  python "$ROOT/bench.py" --vqa --precision fp32 --steps 2 --warmup 1 --no-cpu-baseline
VLB_WGRAD_STREAM=0 one f16_kernel_stats 5 "headline workload on the fp16 build of the library (VLB_PRECISION=f16): same kernels, IEEE fp16 as the 16-bit type" \
  python "$ROOT/bench.py" --precision f16 --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-phase-times --no-clock-probe
