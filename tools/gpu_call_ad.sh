#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 200 python -m pytest tests/test_vision_gpu.py -q -x -k "im2col or stem or resnet" 2>&1 | tail -3
echo "== dp2 e2e"; timeout 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tools/dp2_check.py --e2e > $O/dp2_e2e.log 2>&1; echo "rc=$?"; tail -4 $O/dp2_e2e.log
timeout 200 python bench.py --e2e --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c1-330
bash tools/make_profiles.sh r02 2>&1 | tail -4
