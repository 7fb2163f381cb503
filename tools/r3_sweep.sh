#!/bin/bash
# Runs ON THE GPU BOX: default bench step under a few env-switched GEMM options (same box, interleaved with the default)
one() { python bench.py --no-cpu-baseline --no-phase-times 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['roofline']['achieved'], d['roofline']['gemm_ms_per_step'])"; }
one default
VLB_GEMM_P8_GROUP=1 one p8_group1
VLB_GEMM_P8_GROUP=4 one p8_group4
VLB_GEMM_P8_GROUP=8 one p8_group8
one default
VLB_GEMM_TN8_GROUP=2 one tn8_group2
VLB_GEMM_TN8_GROUP=4 one tn8_group4
VLB_GEMM_P8_KEEPB=0 one keepb0
VLB_GEMM_P8_KEEPB=1 one keepb1
one default
