#!/bin/bash
# Runs ON THE GPU BOX: kernel traces of the per-GPU batches of a strong-scaling run (64 / 32 samples: the 4- / 8-GPU columns)
#   -> gpurun_out/summary/<tag>_kernel_stats_batch{64,32}.txt
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$ROOT/gpurun_out"; TAG="${1:-r03}"
cd /tmp && export TMPDIR=/tmp
mkdir -p "$OUT/summary"
for b in 64 32; do
  name=kernel_stats_batch$b
  VLB_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats -d "$OUT/tr_$name" -o r -- python "$ROOT/bench.py" --global-batch $b --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-phase-times > "$OUT/tr_$name.log" 2>&1
  { echo "# per-GPU batch $b of the global-256 strong-scaling run on ONE MI355X (no communication), weight-gradient stream serialised; 1 warm-up + 3 timed + 1 instrumented step"
    echo "# python bench.py --global-batch $b --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-phase-times   (commit ${VLB_COMMIT:-?})"
    grep '"metric"' "$OUT/tr_$name.log" | tail -1 | sed 's/^/# bench line of the traced run: /'; python "$ROOT/tools/kstats.py" "$OUT/tr_$name" 5 45; } > "$OUT/summary/${TAG}_$name.txt"
  rm -rf "$OUT/tr_$name"
  sed -n 4,34p "$OUT/summary/${TAG}_$name.txt" | cut -c1-150
done
