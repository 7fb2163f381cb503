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

# ---- weight gradient (TN): dW[N_out, K_in] = dY^T X over M = 25856 rows; the library runs the four of a layer as one grouped launch
if "--more" in sys.argv:
    M = 25856
    X = (torch.rand((M, 768), device=d) * 2 - 1).to(torch.bfloat16)
    for n_out, k_in in ((2304, 768), (768, 768), (3072, 768), (768, 3072)):
        dY = (torch.rand((M, n_out), device=d) * 2 - 1).to(torch.bfloat16)
        Xk = X if k_in == 768 else (torch.rand((M, k_in), device=d) * 2 - 1).to(torch.bfloat16)
        out = torch.empty((n_out, k_in), dtype=torch.float32, device=d)
        us_v = t(lambda: torch.matmul(dY.t(), Xk))                 # bf16 output (the vendor call cannot accumulate to fp32 from bf16 inputs here)
        fl = 2.0 * M * n_out * k_in
        print("wgrad %5d x %5d over %d rows : vendor (bf16 out) %7.1f us = %6.1f TF/s" % (n_out, k_in, M, us_v, fl / us_v / 1e6))
    # ---- attention: torch's fused SDPA (flash / CK backend) at the model shape, forward and forward + backward
    import torch.nn.functional as F
    B, nh, S, dh = 256, 12, 101, 64
    q, k, v = [(torch.randn((B, nh, S, dh), device=d)).to(torch.bfloat16).requires_grad_(True) for _ in range(3)]
    us_f = t(lambda: F.scaled_dot_product_attention(q, k, v, dropout_p=0.1))
    def fb():
        o = F.scaled_dot_product_attention(q, k, v, dropout_p=0.1)
        o.backward(torch.ones_like(o))
    us_fb = t(fb)
    print("attention B=256 h=12 S=101 d=64 p=0.1: torch SDPA fwd %.1f us, fwd+bwd %.1f us (vlb: fwd 41, bwd 93 -> 134)" % (us_f, us_fb))
