#!/usr/bin/env python
"""LayerNorm forward/backward bandwidth on the encoder shape (rows = B*101, H = 768)."""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("vl-bert_amd.ops")

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
a = ap.parse_args()
rows, H, d = a.batch * 101, 768, "cuda:0"
x = torch.randn((rows, H), device=d).to(torch.bfloat16)
dy = torch.randn((rows, H), device=d).to(torch.bfloat16)
gamma, beta = torch.ones(H, device=d), torch.zeros(H, device=d)
y, dx, dd = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
stats = torch.empty((rows, 2), device=d)
dg, db = torch.zeros(H, device=d), torch.zeros(H, device=d)
seed = torch.zeros(1, dtype=torch.int32, device=d)
ws = torch.zeros(ops.ln_bwd_workspace_floats(H), device=d)


def timeit(f, n=30):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


mb = rows * H * 2 / 1e6
t = timeit(lambda: ops.layernorm_fwd(x, gamma, beta, y, stats))
print("ln fwd             %7.1f us  %6.0f GB/s" % (t, 2 * mb / t * 1e3))
t = timeit(lambda: ops.layernorm_bwd(dy, x, stats, gamma, dx=dx, dx_drop=dd, drop_p=0.1, seed=seed, tag=3, dgamma=dg, dbeta=db))
print("ln bwd dx+drop+gb  %7.1f us  %6.0f GB/s" % (t, 4 * mb / t * 1e3))
t = timeit(lambda: ops.layernorm_bwd(dy, x, stats, gamma, dx=dx, dx_drop=dd, drop_p=0.1, seed=seed, tag=3, dgamma=dg, dbeta=db, workspace=ws))
print("  .. with workspace %6.1f us  %6.0f GB/s" % (t, 4 * mb / t * 1e3))
t = timeit(lambda: ops.layernorm_bwd(dy, x, stats, gamma, dx=dx, dx_drop=dd, drop_p=0.1, seed=seed, tag=3))
print("ln bwd dx+drop     %7.1f us  %6.0f GB/s" % (t, 4 * mb / t * 1e3))
t = timeit(lambda: ops.layernorm_bwd(dy, x, stats, gamma, dx=dx, dgamma=dg, dbeta=db))
print("ln bwd dx+gb       %7.1f us  %6.0f GB/s" % (t, 3 * mb / t * 1e3))
