#!/usr/bin/env python
"""Per-shape throughput of vlb_gemm_nt_bf16 on the GEMM shapes of one VL-BERT-base training step
(B samples per GPU, S=101).  Random uniform [-1,1) operands (zero-filled data flatters the clock)."""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("vl-bert_amd.ops")


def bench(M, N, K, mode, iters=20):
    d = "cuda:0"
    A = (torch.rand((M, K), device=d) * 2 - 1).to(torch.bfloat16)
    B = (torch.rand((N, K), device=d) * 2 - 1).to(torch.bfloat16)
    kw = {}
    ldc = (N + 63) // 64 * 64
    if mode == "wgrad":
        C = torch.zeros((M, ldc), dtype=torch.float32, device=d)[:, :N]
        kw = dict(out_mode=ops.OUT_F32_ATOMIC)
    else:
        C = torch.empty((M, ldc), dtype=torch.bfloat16, device=d)[:, :N]
        if mode == "gelu":
            kw = dict(bias=torch.zeros(N, device=d), act=ops.ACT_GELU, pre=torch.empty((M, ldc), dtype=torch.bfloat16, device=d)[:, :N])
        elif mode == "res":
            kw = dict(bias=torch.zeros(N, device=d), res=torch.zeros((M, ldc), dtype=torch.bfloat16, device=d)[:, :N])
        elif mode == "bias":
            kw = dict(bias=torch.zeros(N, device=d))
    if mode == "wgrad":
        work = torch.empty(max(ops.wgrad_workspace_floats(M, N, K), 4), dtype=torch.float32, device=d)
        run = lambda: ops.wgrad_nt(A, B, C, workspace=work)
    else:
        run = lambda: ops.gemm_nt(A, B, C, **kw)
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, 2.0 * M * N * K / ms / 1e9


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    a = ap.parse_args()
    M = a.batch * 101
    Mp = (M + 63) // 64 * 64
    BT = a.batch * 64
    shapes = [
        ("qkv fwd", M, 2304, 768, "bias"), ("attn-out fwd", M, 768, 768, "res"), ("ffn1 fwd", M, 3072, 768, "gelu"),
        ("ffn2 fwd", M, 768, 3072, "res"), ("qkv dgrad", M, 768, 2304, "res"), ("ffn1 dgrad", M, 768, 3072, "res"),
        ("ffn2 dgrad", M, 3072, 768, "plain"), ("qkv wgrad", 2304, 768, Mp, "wgrad"), ("out wgrad", 768, 768, Mp, "wgrad"),
        ("ffn1 wgrad", 3072, 768, Mp, "wgrad"), ("ffn2 wgrad", 768, 3072, Mp, "wgrad"),
        ("decoder fwd", BT, 30522, 768, "bias"), ("decoder dgrad", BT, 768, 30528, "plain"), ("decoder wgrad", 30522, 768, BT, "wgrad"),
        ("square 4096", 4096, 4096, 4096, "plain"), ("square 8192", 8192, 8192, 8192, "plain"),
    ]
    print("%-16s %8s %8s %8s  %9s %9s" % ("gemm", "M", "N", "K", "ms", "TFLOP/s"))
    for name, m, n, k, mode in shapes:
        ms, tf = bench(m, n, k, mode)
        print("%-16s %8d %8d %8d  %9.3f %9.1f" % (name, m, n, k, ms, tf))
