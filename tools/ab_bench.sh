#!/bin/bash
# Runs ON THE GPU BOX: the default bench step, library variants interleaved (same box, same minute):
#   tools/ab_bench.sh [variant dir under vl-bert_amd/csrc, default ab0] [rounds, default 2]
V="${1:-ab0}"; R="${2:-2}"
for i in $(seq 1 "$R"); do
  for v in tree "$V"; do
    if [ "$v" = tree ]; then unset VLB_LIB_PATH; else export VLB_LIB_PATH=$PWD/vl-bert_amd/csrc/$v/libvlbert_hip.so; fi
    python bench.py --no-cpu-baseline --no-phase-times 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['roofline']['achieved'], d['roofline']['gemm_ms_per_step'], d['loss'])"
  done
done
