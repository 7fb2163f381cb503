#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_vision_gpu.py -q -x 2>&1 | tail -4
AB=$PWD/vl-bert_amd/csrc/ab/libvlbert_hip.so
for b in 256 32; do for u in old new old new; do
  if [ $u = old ]; then export VLB_LIB_PATH=$AB; else unset VLB_LIB_PATH; fi
  timeout 200 python bench.py --global-batch $b --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('batch $b hash $u ms_per_step', j['ms_per_step'], 'gemm frac', j['roofline']['frac'])"
done; done
unset VLB_LIB_PATH
