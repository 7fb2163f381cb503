"""A/B timing of the encoder's NT GEMM shapes (launcher's own kernel choice) for the library VLB_LIB_PATH points at; a chained pair
FFN1 -> FFN2 (the consumer reads what the producer wrote) is timed as well.  Usage: [VLB_LIB_PATH=...] python tools/p8_ab.py [batch]"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("vl-bert_amd.ops")
D, BF = "cuda:0", torch.bfloat16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
M = B * 101
g = torch.Generator().manual_seed(0)
rnd = lambda *s: ((torch.rand(s, generator=g) * 2 - 1)).to(BF).to(D)
zb = lambda *s: torch.zeros(s, dtype=BF, device=D)
seed = torch.tensor([77], dtype=torch.int32, device=D)
X, Wqkv, QKV = rnd(M, 768), rnd(2304, 768), zb(M, 2304)
Wo, W1, W2 = rnd(768, 768), rnd(3072, 768), rnd(768, 3072)
Z = torch.zeros((M, 768), dtype=torch.float16, device=D)
Zp = (torch.rand((M, 768), generator=g) * 2).half().to(D)
st = torch.cat((torch.zeros(M, 1), torch.ones(M, 1)), 1).contiguous().to(D)
gam, bet, b768, b3072, b2304 = torch.ones(768, device=D), torch.zeros(768, device=D), torch.zeros(768, device=D), torch.zeros(3072, device=D), torch.zeros(2304, device=D)
G, dG, dU, dX = zb(M, 3072), zb(M, 3072), zb(M, 3072), zb(M, 768)
res = rnd(M, 768)
cases = [
    ("qkv fwd (bias)", lambda: ops.gemm_nt(X, Wqkv, QKV, bias=b2304), 2.0 * M * 2304 * 768),
    ("attn-out fwd (drop+LNres)", lambda: ops.gemm_nt(X, Wo, Z, bias=b768, res=Zp, res_ln=(st, gam, bet), drop_p=0.1, seed=seed, tag=1), 2.0 * M * 768 * 768),
    ("ffn1 fwd (gelu+gelu')", lambda: ops.gemm_nt(X, W1, G, bias=b3072, act=ops.ACT_GELU_D, pre=dG), 2.0 * M * 3072 * 768),
    ("ffn2 fwd (drop+LNres)", lambda: ops.gemm_nt(G, W2, Z, bias=b768, res=Zp, res_ln=(st, gam, bet), drop_p=0.1, seed=seed, tag=2), 2.0 * M * 768 * 3072),
    ("ffn2 dgrad (x gelu')", lambda: ops.gemm_nt(X, W1, dU, act=ops.ACT_MULAUX, aux=dG), 2.0 * M * 3072 * 768),
    ("ffn1 dgrad (+res)", lambda: ops.gemm_nt(dU, W2, dX, res=res), 2.0 * M * 768 * 3072),
    ("qkv dgrad (+res)", lambda: ops.gemm_nt(QKV, rnd(768, 2304) if False else WT, dX, res=res), 2.0 * M * 768 * 2304),
    ("ffn1->ffn2 chained", lambda: (ops.gemm_nt(X, W1, G, bias=b3072, act=ops.ACT_GELU_D, pre=dG),
                                    ops.gemm_nt(G, W2, Z, bias=b768, res=Zp, res_ln=(st, gam, bet), drop_p=0.1, seed=seed, tag=2)), 4.0 * M * 768 * 3072),
]
WT = rnd(768, 2304)
tot = 0.0
print("library:", os.environ.get("VLB_LIB_PATH", "default"), " batch", B)
for name, fn, fl in cases:
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    if "chained" not in name:
        tot += us
    print("%-28s %8.1f us %7.0f TFLOP/s" % (name, us, fl / us / 1e6), flush=True)
print("%-28s %8.1f us" % ("sum of the 7 shapes", tot))
