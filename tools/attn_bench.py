#!/usr/bin/env python
"""Attention forward/backward kernel timing at the model shape (B per GPU x 12 heads, S=101)."""
import argparse, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("vl-bert_amd.ops")
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=256); ap.add_argument("--p", type=float, nargs="+", default=[0.1]); ap.add_argument("--seq", type=int, nargs="+", default=[101])
a = ap.parse_args()
d = "cuda:0"
def run(S, P):
  B, H, nh = a.batch, 768, 12
  qkv = (torch.randn((B * S, 3 * H), device=d)).to(torch.bfloat16)
  mask = torch.ones((B, S), device=d)
  ctx = torch.zeros((B * S, H), dtype=torch.bfloat16, device=d)
  lse = torch.zeros((B, nh, S), device=d)
  dctx = (torch.randn((B * S, H), device=d)).to(torch.bfloat16)
  dqkv = torch.zeros_like(qkv)
  seed = torch.tensor([123], dtype=torch.int32, device=d)
  def t(fn, iters=10):
      for _ in range(2): fn()
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      for _ in range(iters): fn()
      e1.record(); torch.cuda.synchronize()
      return e0.elapsed_time(e1) / iters * 1e3
  f = t(lambda: ops.attention_fwd(qkv, mask, ctx, lse, B, S, H, nh, drop_p=P, seed=seed, tag=1))
  b = t(lambda: ops.attention_bwd(qkv, mask, ctx, lse, dctx, dqkv, B, S, H, nh, drop_p=P, seed=seed, tag=1))
  fl = 4.0 * S * S * 64 * B * nh
  print("attention B=%d S=%d p=%.2f: fwd %.1f us (%.1f TF/s)  bwd %.1f us (%.1f TF/s, 2.5x fwd flops)" % (B, S, P, f, fl / f / 1e6, b, 2.5 * fl / b / 1e6))

for S_ in a.seq:
  for P_ in a.p:
    run(S_, P_)
