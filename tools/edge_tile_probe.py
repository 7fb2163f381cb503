#!/usr/bin/env python
"""How much does the EDGE-tile epilogue of the small-tile GEMM kernels cost?  Times vlb_gemm_nt_bf16 with the
bias + residual epilogue (the dgrad form of the N = 768 GEMMs) at the row counts of a 32- / 64-sample rank (M = 3232 / 6464: one partial
128-row tile) against the next multiple of 128 (no partial tile, 3 % / 1.5 % MORE work).   python tools/edge_tile_probe.py"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("vl-bert_amd.ops")
d = "cuda:0"
def t(fn, iters=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for N, K in ((768, 768), (768, 2304), (768, 3072), (3072, 768)):
    for M0 in (3232, 6464):
        row = []
        for M in (M0, (M0 + 127) // 128 * 128):
            A = (torch.rand((M, K), device=d) * 2 - 1).to(torch.bfloat16)
            B = (torch.rand((N, K), device=d) * 2 - 1).to(torch.bfloat16)
            R = (torch.rand((M, N), device=d) * 2 - 1).to(torch.bfloat16)
            C = torch.empty((M, N), dtype=torch.bfloat16, device=d)
            bias = torch.zeros((N,), device=d)
            row.append((M, t(lambda: ops.gemm_nt(A, B, C, bias=bias, res=R))))
        print("N=%4d K=%4d : M=%d %.1f us | M=%d %.1f us  (edge-tile penalty %.1f us)" % (N, K, row[0][0], row[0][1], row[1][0], row[1][1], row[0][1] - row[1][1]))
