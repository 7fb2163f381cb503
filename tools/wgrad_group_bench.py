#!/usr/bin/env python
"""The grouped weight-gradient launch of one encoder layer (vlb_wgrad_tn_group_bf16: QKV, attention output, FFN1, FFN2 over the same
R rows) with and without the bias-gradient column sums, and per member.  Measured in BOTH orders: the first timed loops of a fresh process
run 10-20 % slow (clock / power ramp), which once read as "the column sums cost 16 %" -- in steady state they cost ~3 %.
python tools/wgrad_group_bench.py [batch]"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("vl-bert_amd.ops")
d = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
R = B * 101


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


def member(Mo, No):
    dy = (torch.rand((R, Mo), device=d) * 2 - 1).to(torch.bfloat16)
    x = (torch.rand((R, No), device=d) * 2 - 1).to(torch.bfloat16)
    return dy, x, torch.zeros((Mo, No), device=d), torch.zeros(Mo, device=d)


shapes = [("qkv", 2304, 768), ("out", 768, 768), ("ffn1", 3072, 768), ("ffn2", 768, 3072)]
M = [member(mo, no) for _, mo, no in shapes]
ws = torch.empty(2 * sum(mo * no for _, mo, no in shapes) + 1024, device=d)
flops = 2.0 * R * sum(mo * no for _, mo, no in shapes)
with_cs, without = [(a, b, c, s) for a, b, c, s in M], [(a, b, c, None) for a, b, c, s in M]
for label, items in (("without", without), ("with column sums", with_cs), ("without", without), ("with column sums", with_cs)):      # (both orders)
    us = t(lambda: ops.wgrad_tn_group(items, workspace=ws, accumulate=False))
    print("layer group, batch %d, %-17s: %7.1f us  %7.1f TFLOP/s" % (B, label, us, flops / us / 1e6))
for (name, mo, no), (a, b, c, s) in zip(shapes, M):
    for label, cs in (("with", s), ("without", None)):
        us = t(lambda: ops.wgrad_tn_group([(a, b, c, cs)], workspace=ws, accumulate=False))
        print("  %-5s alone %-8s: %7.1f us  %7.1f TFLOP/s" % (name, label, us, 2.0 * R * mo * no / us / 1e6))
