#!/usr/bin/env python
"""Cost of the fused epilogues on the FFN1 shape (M = B*101, N = 3072, K = 768) and the FFN2-dgrad shape."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("vl-bert_amd.ops")
d = "cuda:0"
M, N, K = 256 * 101, 3072, 768
A = (torch.rand((M, K), device=d) * 2 - 1).to(torch.bfloat16)
B = ((torch.rand((N, K), device=d) * 2 - 1) * 0.05).to(torch.bfloat16)
C = torch.empty((M, N), dtype=torch.bfloat16, device=d)
P = torch.empty_like(C)
X = torch.randn((M, N), device=d).to(torch.bfloat16)
bias = torch.zeros(N, device=d)
seed = torch.zeros(1, dtype=torch.int32, device=d)


def t(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for name, kw in [("plain", {}), ("bias", dict(bias=bias)), ("relu", dict(bias=bias, act=ops.ACT_RELU)),
                 ("gelu (no pre)", dict(bias=bias, act=ops.ACT_GELU)), ("gelu + pre", dict(bias=bias, act=ops.ACT_GELU, pre=P)),
                 ("gelu_d (no pre)", dict(bias=bias, act=ops.ACT_GELU_D)), ("gelu_d + pre", dict(bias=bias, act=ops.ACT_GELU_D, pre=P)),
                 ("x aux", dict(act=ops.ACT_MULAUX, aux=X)), ("x gelu'(aux)", dict(act=ops.ACT_DGELU, aux=X)),
                 ("bias+res", dict(bias=bias, res=X)), ("bias+drop+res", dict(bias=bias, res=X, drop_p=0.1, seed=seed, tag=3))]:
    us = t(lambda: ops.gemm_nt(A, B, C, **kw))
    print("%-18s %7.1f us  %6.0f TF/s" % (name, us, 2.0 * M * N * K / us / 1e6))
