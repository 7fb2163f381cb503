#!/bin/bash
# Runs ON THE GPU BOX: the bench lines quoted in DESIGN.md / README.md (round 3) -> gpurun_out/r3num/*.json
set -u
OUT=gpurun_out/r3num; mkdir -p $OUT
run() { name=$1; shift; python bench.py "$@" > $OUT/$name.json 2> $OUT/$name.err; python - <<P
import json
try:
    d = json.load(open("$OUT/$name.json"))
    r = d.get("roofline", {})
    print("%-14s %9.3f ms %10.1f samples/s  gemm %7.1f TF frac %.3f  %s" % ("$name", d["ms_per_step"], d["value"], r.get("achieved", 0), r.get("frac", 0), d.get("cpu_baseline", {}).get("value", "")))
except Exception as e:
    print("$name FAILED", e)
P
}
run default --cpu-baseline-full
run b128 --global-batch 128 --no-cpu-baseline --no-phase-times
run b64 --global-batch 64 --no-cpu-baseline --no-phase-times
run b32 --global-batch 32 --no-cpu-baseline --no-phase-times
run f16 --precision f16 --no-cpu-baseline
run e2e --e2e --no-cpu-baseline
run large --large --no-cpu-baseline
run large_f16 --large --precision f16 --no-cpu-baseline
run vqa --vqa --steps 5 --warmup 2 --no-cpu-baseline
run vqa_f16 --vqa --precision f16 --steps 5 --warmup 2 --no-cpu-baseline
run vqa_fp32 --vqa --precision fp32 --steps 3 --warmup 1
run vcr --vcr --steps 3 --warmup 1 --no-cpu-baseline
run vcr_f16 --vcr --precision f16 --steps 3 --warmup 1 --no-cpu-baseline
run g32 --global-batch 32 --graph --no-cpu-baseline --no-phase-times
run g256 --graph --no-cpu-baseline --no-phase-times
