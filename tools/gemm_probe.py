#!/usr/bin/env python
"""A few plain GEMM shapes, few iterations: meant to be run under rocprofv3 --pmc (VLB_GEMM_V3 selects the kernel)."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("vl-bert_amd.ops")
d = "cuda:0"
for M, N, K in [(8192, 8192, 8192), (25856, 3072, 768), (25856, 768, 3072), (16384, 30522, 768)]:
    A = (torch.rand((M, K), device=d) * 2 - 1).to(torch.bfloat16)
    B = (torch.rand((N, K), device=d) * 2 - 1).to(torch.bfloat16)
    ldc = (N + 63) // 64 * 64
    C = torch.empty((M, ldc), dtype=torch.bfloat16, device=d)[:, :N]
    for _ in range(4):
        ops.gemm_nt(A, B, C)
    torch.cuda.synchronize()
