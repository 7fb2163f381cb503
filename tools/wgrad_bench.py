#!/usr/bin/env python
"""TN (LDS transpose reads) vs NT (+ explicit transposes) weight-gradient paths at the model's shapes."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("vl-bert_amd.ops")
d = "cuda:0"

def t(fn, iters=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

print("%-14s %7s %6s %6s  %9s %9s %9s" % ("wgrad", "R", "Mo", "No", "TN ms", "NT+T ms", "TN TF/s"))
for name, R, Mo, No in [("qkv", 25856, 2304, 768), ("out", 25856, 768, 768), ("ffn1", 25856, 3072, 768), ("ffn2", 25856, 768, 3072),
                        ("decoder", 16384, 30522, 768), ("mvrc cls", 9216, 1601, 768), ("downsample", 9216, 768, 4096)]:
    lda = (Mo + 63) // 64 * 64
    dy = ((torch.rand((R, lda), device=d) * 2 - 1).to(torch.bfloat16))[:, :Mo]
    x = (torch.rand((R, No), device=d) * 2 - 1).to(torch.bfloat16)
    C = torch.zeros((Mo, No), dtype=torch.float32, device=d)
    db = torch.zeros(Mo, dtype=torch.float32, device=d)
    Rp = (R + 63) // 64 * 64
    ws = torch.empty(max(ops.wgrad_workspace_floats(Mo, No, Rp), 4), dtype=torch.float32, device=d)
    tg = torch.zeros((Mo, Rp), dtype=torch.bfloat16, device=d)
    ta = torch.zeros((No, Rp), dtype=torch.bfloat16, device=d)
    tn = t(lambda: ops.wgrad_tn(dy, x, C, colsum=db, workspace=ws))
    def nt():
        ops.transpose(dy, tg, colsum=db); ops.transpose(x, ta); ops.wgrad_nt(tg, ta, C, workspace=ws)
    ntt = t(nt)
    print("%-14s %7d %6d %6d  %9.3f %9.3f %9.1f" % (name, R, Mo, No, tn, ntt, 2.0 * R * Mo * No / tn / 1e9))
