#!/usr/bin/env python
"""Where does the epilogue of the large-tile NT GEMM spend its time?  The FFN shape (M = 25856, N = 3072) at K = 128 (the shortest K
loop the kernel takes: the launch is nearly all epilogue) and K = 768, per fused form, with the kernel's timing ablations
(p8_ablate 1: no epilogue, 2: epilogue without its global stores).  Usage: python tools/epi_probe.py [K ...]"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("vl-bert_amd.ops")
lib = importlib.import_module("vl-bert_amd._lib")
d = "cuda:0"
M, N = 256 * 101, int(os.environ.get("EPI_N", 3072))
Ks = [int(a) for a in sys.argv[1:]] or [128, 768]
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


forms = [("bias", dict(bias=bias)), ("gelu_d + pre", dict(bias=bias, act=ops.ACT_GELU_D, pre=P)), ("x aux", dict(act=ops.ACT_MULAUX, aux=X)),
         ("bias+res", dict(bias=bias, res=X)), ("bias+drop+res", dict(bias=bias, res=X, drop_p=0.1, seed=seed, tag=3))]
lib.gemm_set_option("p8_mode", int(os.environ.get("EPI_TILE", 4)))
for K in Ks:
    A = (torch.rand((M, K), device=d) * 2 - 1).to(torch.bfloat16)
    B = ((torch.rand((N, K), device=d) * 2 - 1) * 0.05).to(torch.bfloat16)
    lib.gemm_set_option("p8_ablate", 1)
    base = t(lambda: ops.gemm_nt(A, B, C, bias=bias))
    print("K = %d  N = %d: main loop only %.1f us" % (K, N, base))
    for name, kw in forms:
        r = []
        for ab in (0, 2):
            lib.gemm_set_option("p8_ablate", ab)
            r.append(t(lambda: ops.gemm_nt(A, B, C, **kw)))
        print("  %-16s full %7.1f us   without stores %7.1f us   epilogue = +%.1f us" % (name, r[0], r[1], r[0] - base), flush=True)
    lib.gemm_set_option("p8_ablate", 0)
