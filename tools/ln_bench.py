#!/usr/bin/env python
"""LayerNorm backward micro-benchmark (run on the GPU box):  VLB_LN_BWD4=0|1|2 python tools/ln_bench.py [rows] [H]"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("vl-bert_amd.ops")
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 25856
H = int(sys.argv[2]) if len(sys.argv) > 2 else 768
d = "cuda:0"
g = torch.Generator().manual_seed(0)
x = (torch.randn(rows, H, generator=g) * 2).half().to(d)
dy = torch.randn(rows, H, generator=g).bfloat16().to(d)
gamma = torch.randn(H, generator=g).to(d)
beta = torch.randn(H, generator=g).to(d)
y = torch.empty(rows, H, dtype=torch.bfloat16, device=d)
stats = torch.empty(rows, 2, device=d)
ops.layernorm_fwd(x, gamma, beta, y, stats)
dx = torch.empty_like(dy)
dxd = torch.empty_like(dy)
dg, db = torch.zeros(H, device=d), torch.zeros(H, device=d)
ws = torch.zeros(ops.ln_bwd_workspace_floats(H), device=d)
seed = torch.tensor([12345], dtype=torch.int32, device=d)
run = lambda: ops.layernorm_bwd(dy, x, stats, gamma, dx=dx, dx_drop=dxd, drop_p=0.1, seed=seed, tag=5, dgamma=dg, dbeta=db, workspace=ws)
for _ in range(3):
    run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
print("VLB_LN_BWD4=%s rows %d H %d: %.1f us per call (incl. finalize), %.2f TB/s on %d MB" %
      (os.environ.get("VLB_LN_BWD4", "default"), rows, H, us, rows * H * 8 / us / 1e6, rows * H * 8 >> 20))
