#!/usr/bin/env python
"""Yardstick only (never on the product path): what the vendor library (hipBLASLt through torch.matmul) reaches on the encoder's GEMM
shapes with RANDOM bf16 operands on this box -- zero-filled operands flatter the clock by ~19 % (MI355X_MICROARCH.md, DVFS) -- beside
vlb_gemm_nt_bf16 (bias-only epilogue) on the same tensors.   python tools/vendor_gemm_yardstick.py"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("vl-bert_amd.ops")
d = "cuda:0"
def t(fn, iters=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for M, N, K in ((25856, 2304, 768), (25856, 768, 768), (25856, 3072, 768), (25856, 768, 3072), (25856, 768, 2304), (8192, 8192, 8192)):
    A = (torch.rand((M, K), device=d) * 2 - 1).to(torch.bfloat16)
    B = (torch.rand((N, K), device=d) * 2 - 1).to(torch.bfloat16)
    C = torch.empty((M, N), dtype=torch.bfloat16, device=d)
    bias = torch.zeros((N,), device=d)
    Z = torch.zeros_like(A)
    us_v = t(lambda: torch.matmul(A, B.t(), out=C))
    us_z = t(lambda: torch.matmul(Z, B.t(), out=C))
    us_o = t(lambda: ops.gemm_nt(A, B, C, bias=bias))
    fl = 2.0 * M * N * K
    print("%6d x %5d x %5d : vendor %7.1f us = %6.1f TF/s (zero-filled A: %6.1f TF/s) | vlb_gemm_nt_bf16 %7.1f us = %6.1f TF/s" %
          (M, N, K, us_v, fl / us_v / 1e6, fl / us_z / 1e6, us_o, fl / us_o / 1e6))
