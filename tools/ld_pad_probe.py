#!/usr/bin/env python
"""Does the leading dimension of the operands matter (L2 channel interleave)?  Times the encoder's GEMM shapes with row strides
that are / are not multiples of 512 B.  Usage: python tools/ld_pad_probe.py"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("vl-bert_amd.ops")
d = "cuda:0"


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for M, N, K in [(25856, 768, 3072), (25856, 3072, 768), (25856, 768, 768), (25856, 2304, 768)]:
    row = []
    for pad_a, pad_b, pad_c in [(0, 0, 0), (64, 0, 0), (0, 64, 0), (64, 64, 0), (64, 64, 64), (32, 32, 32)]:
        A = (torch.rand((M, K + pad_a), device=d) * 2 - 1).to(torch.bfloat16)[:, :K]
        B = (torch.rand((N, K + pad_b), device=d) * 2 - 1).to(torch.bfloat16)[:, :K]
        C = torch.empty((M, N + pad_c), dtype=torch.bfloat16, device=d)[:, :N]
        bias = torch.zeros(N, device=d)
        us = timed(lambda: ops.gemm_nt(A, B, C, bias=bias))
        row.append("padA%d,B%d,C%d: %6.1f us (%4.0f TF)" % (pad_a, pad_b, pad_c, us, 2.0 * M * N * K / us / 1e6))
    print("M=%d N=%d K=%d | " % (M, N, K) + " | ".join(row))
# TN weight gradient: dW[N,K] = dY[M,N]^T X[M,K]
for M, N, K in [(25856, 768, 3072), (25856, 3072, 768), (25856, 2304, 768)]:
    row = []
    ws = torch.empty(max(ops.wgrad_workspace_floats(N, K, M), 4), device=d)
    for pad in (0, 64):
        dY = (torch.rand((M, N + pad), device=d) * 2 - 1).to(torch.bfloat16)[:, :N]
        X = (torch.rand((M, K + pad), device=d) * 2 - 1).to(torch.bfloat16)[:, :K]
        G = torch.zeros((N, K), device=d)
        us = timed(lambda: ops.wgrad_tn(dY, X, G, workspace=ws, accumulate=False))
        row.append("pad%d: %6.1f us (%4.0f TF)" % (pad, us, 2.0 * M * N * K / us / 1e6))
    print("TN dW[%d,%d] rows %d | " % (N, K, M) + " | ".join(row))
