#!/usr/bin/env python
"""Where does a large-tile GEMM launch spend its time?  Measured INSIDE the kernel (run on the GPU box):

    python tools/p8_phase_probe.py [batch] [--spread]      (--spread: per-workgroup entry / exit times -- which workgroups end the launch)

For each of the eight NT GEMMs of an encoder layer (M = batch x 101, the epilogue it carries in the step, the library's own kernel
selection) one launch runs with p8_ablate = 4: wave 0 of every workgroup accumulates the shader cycles (s_memtime) it spends in the K
loops and in the epilogues (incl. the tile-end barriers) and stamps entry / exit with the 100 MHz real-time counter.  Printed per
shape: tiles per workgroup, microseconds per tile in the main loop and in the epilogue (median over the workgroups that own the
maximum number of tiles), the kernel residency, the effective clock.  200 plain launches of the same shape run first (sustained
power state)."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("vl-bert_amd.ops")
lib = importlib.import_module("vl-bert_amd._lib")
D = "cuda:0"
BF = torch.bfloat16


def rnd(*s, seed=0, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.rand(s, generator=g) * 2 - 1) * scale


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    batch = int(args[0]) if args else 256
    M = batch * 101
    shapes = [("qkv fwd", 2304, 768, "bias"), ("attn-out fwd", 768, 768, "lnres"), ("ffn1 fwd", 3072, 768, "gelu"), ("ffn2 fwd", 768, 3072, "lnres"),
              ("out dgrad", 768, 768, "plain"), ("qkv dgrad", 768, 2304, "res"), ("ffn1 dgrad", 768, 3072, "res"), ("ffn2 dgrad", 3072, 768, "mulaux")]
    print("%-13s %6s %5s | tiles/wg | main loop us/tile | epilogue us/tile | residency us | launch us | clock MHz" % ("gemm", "N", "K"))
    for name, N, K, kind in shapes:
        A = rnd(M, K, seed=1).to(BF).to(D)
        B = rnd(N, K, seed=2, scale=0.05).to(BF).to(D)
        C = torch.empty((M, N), dtype=(torch.float16 if kind == "lnres" else BF), device=D)
        kw = {}
        if kind in ("bias", "gelu", "res", "lnres"):
            kw["bias"] = torch.zeros(N, device=D)
        if kind == "gelu":
            kw.update(act=ops.ACT_GELU_D)
        if kind == "res":
            kw["res"] = rnd(M, N, seed=5).to(BF).to(D)
        if kind == "mulaux":
            kw.update(act=ops.ACT_MULAUX, aux=rnd(M, N, seed=6).to(BF).to(D))
        if kind == "lnres":
            z = (rnd(M, N, seed=7) * 2).half().to(D)
            st = torch.stack((z.float().mean(1), 1.0 / torch.sqrt(z.float().var(1, unbiased=False) + 1e-12)), 1).contiguous()
            kw.update(res=z, res_ln=(st, torch.ones(N, device=D), torch.zeros(N, device=D)), drop_p=0.1,
                      seed=torch.tensor([77], dtype=torch.int32, device=D), tag=3)
        pre_real = torch.empty((M, N), dtype=BF, device=D) if kind == "gelu" else None
        run = lambda pre: ops.gemm_nt(A, B, C, pre=pre, **kw)
        for _ in range(200):
            run(pre_real)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run(pre_real)
        e1.record()
        torch.cuda.synchronize()
        launch_us = e0.elapsed_time(e1) / 20 * 1e3
        table = torch.zeros((256, 32), dtype=BF, device=D)
        lib.gemm_set_option("p8_ablate", 4)
        try:
            if kind == "gelu":      # `pre` is the GELU' output there: the stamps travel through `aux` (gemm_p8.hip)
                ops.gemm_nt(A, B, C, pre=pre_real, aux=table, **kw)
            else:
                run(table)
        finally:
            lib.gemm_set_option("p8_ablate", 0)
        torch.cuda.synchronize()
        t = table.view(torch.int64).view(256, 8).cpu()
        t = t[t[:, 3] > 0]
        if t.shape[0] == 0:
            print("%-13s %6d %5d | (not on the large-tile core) launch %.1f us" % (name, N, K, launch_us))
            continue
        tiles = t[:, 6].max()
        full = t[t[:, 6] == tiles]
        clk = ((full[:, 2] - full[:, 0]).double() / (full[:, 3] - full[:, 1]).double() * 100.0).median()      # MHz
        main_us = (full[:, 4].double() / tiles / clk).median()
        epi_us = (full[:, 5].double() / tiles / clk).median()
        res_us = ((full[:, 3] - full[:, 1]).double() / 100.0).median()
        print("%-13s %6d %5d | %8d | %17.2f | %16.2f | %12.1f | %9.1f | %9.0f" % (name, N, K, int(tiles), main_us, epi_us, res_us, launch_us, clk), flush=True)
        if "--spread" in sys.argv:      # which workgroups end the launch?  (100 MHz ticks -> us, relative to the first entry stamp)
            ids = torch.nonzero(table.view(torch.int64).view(256, 8).cpu()[:, 3] > 0).flatten()
            t0 = t[:, 1].min()
            start, end = (t[:, 1] - t0).double() / 100.0, (t[:, 3] - t0).double() / 100.0
            order = torch.argsort(end, descending=True)
            print("    workgroups %d | entry spread %.1f us | exit: median %.1f, max %.1f us | tiles per workgroup: %s" % (
                t.shape[0], float(start.max()), float(end.median()), float(end.max()),
                ", ".join("%d x %d" % (int((t[:, 6] == k).sum()), int(k)) for k in sorted(set(t[:, 6].tolist()), reverse=True))))
            print("    last to exit (block: exit us / tiles / main us / epilogue us): " + ", ".join(
                "%d: %.1f / %d / %.1f / %.1f" % (int(ids[i]), float(end[i]), int(t[i, 6]), float(t[i, 4]) / float(clk), float(t[i, 5]) / float(clk))
                for i in order[:6].tolist()))
            print("    first to exit: " + ", ".join(
                "%d: %.1f / %d / %.1f / %.1f" % (int(ids[i]), float(end[i]), int(t[i, 6]), float(t[i, 4]) / float(clk), float(t[i, 5]) / float(clk))
                for i in order[-4:].tolist()))


if __name__ == "__main__":
    main()
