#!/bin/bash
# Runs ON THE GPU BOX: the other bench modes quoted in DESIGN.md / README.md (the default line and the small batches come from
# tools/r3_verify.sh / r3_ab2.sh) -> gpurun_out/r3num/*.json
set -u
OUT=gpurun_out/r3num; mkdir -p $OUT
run() { name=$1; shift; timeout 300 python bench.py "$@" > $OUT/$name.json 2> $OUT/$name.err; python - <<P
import json
try:
    d = json.load(open("$OUT/$name.json"))
    r = d.get("roofline", {})
    print("%-10s %9.3f ms %10.1f samples/s  gemm %7.1f TF frac %.3f" % ("$name", d["ms_per_step"], d["value"], r.get("achieved", 0), r.get("frac", 0)))
except Exception as e:
    print("$name FAILED", e)
P
}
run f16 --precision f16 --no-cpu-baseline --no-phase-times
run e2e --e2e --no-cpu-baseline --no-phase-times
run large --large --no-cpu-baseline --no-phase-times
run vqa --vqa --steps 5 --warmup 2 --no-cpu-baseline
run vqa_fp32 --vqa --precision fp32 --steps 3 --warmup 1 --no-cpu-baseline
run vcr --vcr --steps 3 --warmup 1 --no-cpu-baseline
