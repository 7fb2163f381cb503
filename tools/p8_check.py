#!/usr/bin/env python
"""Correctness + throughput of the large-tile GEMM core (vl-bert_amd/csrc/gemm_p8.hip) on MI355X.

    python tools/p8_check.py check          every fused epilogue / tile height / edge shape against a torch fp32 reference and against
                                           the 128x128 kernels (same inputs, same dropout seed -> identical masks)
    python tools/p8_check.py bench [B]      per-shape TFLOP/s on the GEMM shapes of one VL-BERT-base step (B samples, S = 101) for the
                                           dispatcher variants: 128x128 kernels | 256-row tiles | 320-row tiles | cost model
Each mode is meant to be run under `timeout` (a scheduling bug in a hand-synchronised kernel shows up as a hang)."""
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
    return ((torch.rand(s, generator=g) * 2 - 1) * scale)


def run_case(M, N, K, epi, mode, seed_t, ldpad=0):
    """returns the bf16 result(s) as fp32 CPU tensors"""
    lib.gemm_set_option("p8_mode", mode)
    A = rnd(M, K, seed=1).to(BF).to(D)
    Bm = rnd(N, K, seed=2).to(BF).to(D)
    ldc = (N + 63) // 64 * 64 + ldpad
    C = torch.full((M, ldc), 7.0, dtype=BF, device=D)[:, :N]
    bias = rnd(N, seed=3).float().to(D)
    side = rnd(M, N, seed=4).to(BF)
    side_g = torch.zeros((M, ldc), dtype=BF, device=D)[:, :N]
    side_g.copy_(side.to(D))
    kw = {}
    pre = None
    if epi == 0:
        kw = dict(bias=bias)
    elif epi == 1:
        pre = torch.full((M, ldc), 5.0, dtype=BF, device=D)[:, :N]
        kw = dict(bias=bias, act=ops.ACT_GELU_D, pre=pre)
    elif epi == 2:
        kw = dict(act=ops.ACT_MULAUX, aux=side_g)
    elif epi == 3:
        kw = dict(bias=bias, res=side_g, drop_p=0.1, seed=seed_t, tag=17)
    elif epi == 4:
        kw = dict(bias=bias, res=side_g)
    elif epi == 5:
        kw = dict(bias=bias, act=ops.ACT_RELU)
    elif epi == 8:       # relu(acc + bias + res): Bottleneck forward tail
        kw = dict(bias=bias, res=side_g, act=ops.ACT_RES_RELU)
    elif epi == 10:      # acc where aux > 0: ReLU backward
        kw = dict(act=ops.ACT_RELU_MASK, aux=side_g)
    elif epi in (6, 7):      # LayerNorm-residual from fp16 rows, fp16 output (epi 6: with dropout)
        z = (side.float() * 2.0 + 0.3).half()
        zg = torch.zeros((M, ldc), dtype=torch.float16, device=D)[:, :N]
        zg.copy_(z.to(D))
        zf = z.float()
        mean = zf.mean(1, keepdim=True)
        rstd = 1.0 / torch.sqrt(zf.var(1, unbiased=False, keepdim=True) + 1e-12)
        st = torch.cat((mean, rstd), 1).contiguous().to(D)
        gam = (1.0 + 0.2 * rnd(N, seed=8)).float().to(D)
        bet = (0.1 * rnd(N, seed=9)).float().to(D)
        C = torch.full((M, ldc), 7.0, dtype=torch.float16, device=D)[:, :N]
        kw = dict(bias=bias, res=zg, res_ln=(st, gam, bet))
        if epi == 6:
            kw.update(drop_p=0.1, seed=seed_t, tag=17)
        side = ((zf - mean) * rstd * gam.cpu() + bet.cpu())      # what the reference adds
    ops.gemm_nt(A, Bm, C, **kw)
    torch.cuda.synchronize()
    out = [C.float().cpu()]
    if pre is not None:
        out.append(pre.float().cpu())
    return out, (A, Bm, bias, side)


def reference(M, N, K, epi, ins):
    A, Bm, bias, side = ins
    acc = A.float().cpu() @ Bm.float().cpu().t()
    b = bias.cpu()
    if epi == 0:
        return [acc + b]
    if epi == 1:
        x = acc + b
        cdf = 0.5 * (1 + torch.erf(x / 2 ** 0.5))
        return [x * cdf, cdf + x * torch.exp(-0.5 * x * x) / (2 * 3.141592653589793) ** 0.5]
    if epi == 2:
        return [acc * side.float()]
    if epi in (4, 7):
        return [acc + b + side.float()]
    if epi == 5:
        return [torch.relu(acc + b)]
    if epi == 8:
        return [torch.relu(acc + b + side.float())]
    if epi == 10:
        return [acc * (side.float() > 0)]
    return None     # dropout: compared with the 128x128 kernel only (same counter RNG)


def check():
    seed_t = torch.tensor([12345], dtype=torch.int32, device=D)
    shapes = [(4096, 2304, 768), (8192 + 256, 768, 768), (6464, 768, 3072), (25856, 768, 768), (5000, 1000, 256), (4100, 3000 + 2, 128),
              (16384, 30522, 768)]
    bad = 0
    for M, N, K in shapes:
        for mode in (3, 4, 5):
            for epi in (0, 1, 2, 3, 4, 5, 6, 7, 8, 10):
                if N > 20000 and epi not in (0,):
                    continue
                lib.gemm_set_option("p8_min_tiles", 1)
                got, ins = run_case(M, N, K, epi, mode, seed_t)
                old, _ = run_case(M, N, K, epi, 0, seed_t)
                ref = reference(M, N, K, epi, ins)
                for k, (g, o) in enumerate(zip(got, old)):
                    scale = float(o.abs().max())
                    d_old = float((g - o).abs().max())
                    d_ref = float((g - ref[k]).abs().max()) if ref is not None else float("nan")
                    # vs old kernel: both round fp32 -> bf16 once, tiny accumulation-order differences only; vs fp32 torch: bf16 rounding
                    ok = d_old <= 2.5e-2 * max(scale, 1e-6) and (ref is None or d_ref <= 1.2e-2 * max(scale, 1e-6)) and bool(torch.isfinite(g).all())
                    frac_diff = float(((g - o).abs() > 1e-2 * max(scale, 1e-6)).float().mean())
                    if epi in (3, 6):   # identical dropout masks: the zero patterns must coincide
                        ok = ok and bool(((g == 0) == (o == 0)).float().mean() > 0.9999)
                    print("M %6d N %6d K %5d tile %d epi %d out %d: |d old| %.3e |d ref| %.3e scale %.3e frac>1%% %.2e %s"
                          % (M, N, K, 64 * mode, epi, k, d_old, d_ref, scale, frac_diff, "ok" if ok else "FAIL"), flush=True)
                    bad += 0 if ok else 1
    # rows beyond ldc / neighbours untouched: padded C columns keep their fill value
    lib.gemm_set_option("p8_mode", 4)
    lib.gemm_set_option("p8_min_tiles", 1)
    A = rnd(3000, 256, seed=1).to(BF).to(D)
    Bm = rnd(1000, 256, seed=2).to(BF).to(D)
    Cfull = torch.full((3000 + 8, 1024), 7.0, dtype=BF, device=D)
    ops.gemm_nt(A, Bm, Cfull[:3000, :1000])
    torch.cuda.synchronize()
    ok = bool((Cfull[:3000, 1000:] == 7.0).all()) and bool((Cfull[3000:] == 7.0).all())
    print("padding columns / rows beyond M untouched: %s" % ("ok" if ok else "FAIL"))
    bad += 0 if ok else 1
    print("P8 CHECK %s (%d failures)" % ("PASSED" if bad == 0 else "FAILED", bad), flush=True)
    return bad


def bench_one(M, N, K, mode_kw, opt, iters=20):
    for k, v in opt.items():
        lib.gemm_set_option(k, v)
    A = rnd(M, K, seed=1).to(BF).to(D)
    Bm = rnd(N, K, seed=2).to(BF).to(D)
    ldc = (N + 63) // 64 * 64
    C = torch.empty((M, ldc), dtype=BF, device=D)[:, :N]
    kw = {}
    if mode_kw == "bias":
        kw = dict(bias=torch.zeros(N, device=D))
    elif mode_kw == "gelu":
        kw = dict(bias=torch.zeros(N, device=D), act=ops.ACT_GELU_D, pre=torch.empty((M, ldc), dtype=BF, device=D)[:, :N])
    elif mode_kw == "dropres":
        kw = dict(bias=torch.zeros(N, device=D), res=rnd(M, N, seed=5).to(BF).to(D), drop_p=0.1, seed=torch.tensor([77], dtype=torch.int32, device=D), tag=3)
    elif mode_kw == "res":
        kw = dict(bias=torch.zeros(N, device=D), res=rnd(M, N, seed=5).to(BF).to(D))
    elif mode_kw == "mulaux":
        kw = dict(act=ops.ACT_MULAUX, aux=rnd(M, N, seed=6).to(BF).to(D))
    run = lambda: ops.gemm_nt(A, Bm, C, **kw)
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


def bench(batch):
    M = batch * 101
    BT = batch * 64
    shapes = [("qkv fwd", M, 2304, 768, "bias"), ("attn-out fwd", M, 768, 768, "dropres"), ("ffn1 fwd", M, 3072, 768, "gelu"),
              ("ffn2 fwd", M, 768, 3072, "dropres"), ("out dgrad", M, 768, 768, "plain"), ("qkv dgrad", M, 768, 2304, "res"),
              ("ffn1 dgrad", M, 768, 3072, "res"), ("ffn2 dgrad", M, 3072, 768, "mulaux"), ("decoder fwd", BT, 30522, 768, "bias"),
              ("square 4096", 4096, 4096, 4096, "plain"), ("square 8192", 8192, 8192, 8192, "plain")]
    variants = [("128x128", dict(p8_mode=0)), ("p8 192", dict(p8_mode=3, p8_group=2, p8_min_tiles=1)), ("p8 256", dict(p8_mode=4, p8_keepb=1, p8_group=2, p8_min_tiles=1)),
                ("p8 320", dict(p8_mode=5, p8_group=2, p8_min_tiles=1)), ("p8 model", dict(p8_mode=1, p8_keepb=1, p8_group=2, p8_min_tiles=160))]
    print("%-14s %7s %6s %6s | " % ("gemm", "M", "N", "K") + " | ".join("%-13s" % v[0] for v in variants))
    tot = [0.0] * len(variants)
    for name, m, n, k, kwm in shapes:
        row = []
        for vi, (vn, opt) in enumerate(variants):
            ms, tf = bench_one(m, n, k, kwm, opt)
            row.append("%6.1fus %5.0f" % (ms * 1e3, tf))
            if not name.startswith("square"):
                tot[vi] += ms * (1 if name.startswith("decoder") else 12)
        print("%-14s %7d %6d %6d | " % (name, m, n, k) + " | ".join(row), flush=True)
    print("%-36s | " % "NT ms per step (12 layers + decoder fwd)" + " | ".join("%13.2f" % t for t in tot), flush=True)


def bench_model(batch):
    """the step's NT GEMM shapes with the library's OWN selection and current options (environment / VLB_LIB_PATH decide the variant):
    one column, more iterations -- for A/B runs of compile-time or env-switched variants (tools/gpu_call.sh gemm:<variants>)"""
    M = batch * 101
    shapes = [("qkv fwd", M, 2304, 768, "bias"), ("attn-out fwd", M, 768, 768, "dropres"), ("ffn1 fwd", M, 3072, 768, "gelu"),
              ("ffn2 fwd", M, 768, 3072, "dropres"), ("out dgrad", M, 768, 768, "plain"), ("qkv dgrad", M, 768, 2304, "res"),
              ("ffn1 dgrad", M, 768, 3072, "res"), ("ffn2 dgrad", M, 3072, 768, "mulaux")]
    tot = 0.0
    for rep in range(2):
        tot = 0.0
        for name, m, n, k, kwm in shapes:
            ms, tf = bench_one(m, n, k, kwm, {}, iters=40)
            tot += ms
            if rep == 1:
                print("%-14s %7d %6d %6d | %7.1f us %6.0f TFLOP/s" % (name, m, n, k, ms * 1e3, tf), flush=True)
    print("NT us per layer (8 GEMMs): %.1f" % (tot * 1e3), flush=True)


def bench_wgrad(batch):
    """weight-gradient (TN) shapes of one step: 128x128 TN kernel vs the large-tile core"""
    M = batch * 101
    shapes = [("qkv wgrad", M, 2304, 768), ("out wgrad", M, 768, 768), ("ffn1 wgrad", M, 3072, 768), ("ffn2 wgrad", M, 768, 3072),
              ("decoder wgrad (compact)", 4096, 30522, 768), ("decoder wgrad (full)", batch * 64, 30522, 768)]
    print("%-26s %7s %6s %6s | %-16s | %-16s" % ("wgrad", "R", "Mo", "No", "128x128 TN", "tn8"))
    tot = [0.0, 0.0]
    for name, R, Mo, No in shapes:
        dY = torch.zeros((R, (Mo + 63) // 64 * 64), dtype=BF, device=D)[:, :Mo]
        dY.copy_(rnd(R, Mo, seed=1).to(BF).to(D))
        X = rnd(R, No, seed=2).to(BF).to(D)
        C = torch.zeros((Mo, No), dtype=torch.float32, device=D)
        db = torch.zeros(Mo, dtype=torch.float32, device=D)
        work = torch.empty(max(ops.wgrad_workspace_floats(Mo, No, R), 4), dtype=torch.float32, device=D)
        row = []
        for vi, mode in enumerate((0, 1)):
            lib.gemm_set_option("tn8_mode", mode)
            run = lambda: ops.wgrad_tn(dY, X, C, colsum=db, workspace=work, accumulate=False)
            for _ in range(3):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            row.append("%7.1fus %6.0f TF" % (ms * 1e3, 2.0 * R * Mo * No / ms / 1e9))
            if not name.startswith("decoder wgrad (full"):
                tot[vi] += ms * (1 if name.startswith("decoder") else 12)
        print("%-26s %7d %6d %6d | " % (name, R, Mo, No) + " | ".join(row), flush=True)
    print("TN ms per step (12 layers + compact decoder): %.2f | %.2f" % tuple(tot), flush=True)


def bench_wgrad_vision():
    """layer3 / layer4 1x1-convolution weight gradients of the e2e trunk (8 images): R = 19152 rows is not a multiple of 128, so the
    large-tile core is out; would zero-padding the rows to 19200 pay?  128x128 TN kernel at R = 19152 vs both kernels at 19200"""
    for name, R, Mo, No in (("layer3 conv1", 19152, 256, 1024), ("layer3 conv3", 19152, 1024, 256), ("layer3 down", 19152, 1024, 512),
                            ("layer4 conv1", 56448, 512, 2048), ("layer4 conv3", 56448, 2048, 512)):
        row = []
        for Rp, mode in ((R, 0), ((R + 127) // 128 * 128, 0), ((R + 127) // 128 * 128, 1)):
            dY = torch.zeros((Rp, Mo), dtype=BF, device=D); dY[:R].copy_(rnd(R, Mo, seed=1).to(BF).to(D))
            X = torch.zeros((Rp, No), dtype=BF, device=D); X[:R].copy_(rnd(R, No, seed=2).to(BF).to(D))
            C = torch.zeros((Mo, No), dtype=torch.float32, device=D)
            sc = torch.ones(Mo, dtype=torch.float32, device=D)
            work = torch.empty(64 * Mo * No + 64, dtype=torch.float32, device=D)
            lib.gemm_set_option("tn8_mode", mode)
            run = lambda: ops.wgrad_tn_rowscale(dY, X, C, sc, work, accumulate=True)
            for _ in range(3):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            row.append("R %5d tn8 %d: %6.1f us %4.0f TF" % (Rp, mode, ms * 1e3, 2.0 * R * Mo * No / ms / 1e9))
        print("%-14s %5d x %5d | " % (name, Mo, No) + " | ".join(row), flush=True)
    lib.gemm_set_option("tn8_mode", 1)


def bench_wgrad_group(batch):
    """grouped per-layer weight gradient (4 gradients, one launch): equal 2-slice cut vs the uneven 3-slice cut"""
    R = (batch * 101 + 127) // 128 * 128
    H, I = 768, 3072
    shapes = [(H, I), (I, H), (H, H), (3 * H, H)]
    items = []
    for k, (Mo, No) in enumerate(shapes):
        items.append((rnd(R, Mo, seed=30 + k).to(BF).to(D), rnd(R, No, seed=40 + k).to(BF).to(D),
                      torch.zeros((Mo, No), dtype=torch.float32, device=D), torch.zeros(Mo, dtype=torch.float32, device=D)))
    work = torch.empty(3 * sum(a * b for a, b in shapes) + 64, dtype=torch.float32, device=D)
    flops = 2.0 * R * sum(a * b for a, b in shapes)
    outs = []
    for uneven in (0, 1, 0, 1):
        lib.gemm_set_option("tn8_uneven", uneven)
        run = lambda: ops.wgrad_tn_group(items, workspace=work, accumulate=False)
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        outs.append([t[2].clone() for t in items])
        print("grouped wgrad batch %d R %d uneven %d: %7.1f us %6.0f TFLOP/s (incl. slab reduce)" % (batch, R, uneven, ms * 1e3, flops / ms / 1e9),
              flush=True)
    for a, b in zip(outs[0], outs[1]):
        print("  |uneven - equal| max %.3e scale %.3e" % (float((a - b).abs().max()), float(a.abs().max())))
    lib.gemm_set_option("tn8_uneven", 0)


def check_ring():
    """ring kernels (nt_ring 2: 128x128 / 4 stages, 3: 128x64 / 3 stages) against the two-stage kernel and torch fp32"""
    seed_t = torch.tensor([12345], dtype=torch.int32, device=D)
    shapes = [(3232, 768, 768), (3232, 3072, 768), (3232, 768, 3072), (6464, 2304, 768), (1000, 1000, 256), (777, 130, 64),
              (4100, 3000 + 2, 128), (128, 64, 64), (25856, 768, 768)]
    bad = 0
    lib.gemm_set_option("p8_mode", 0)
    for M, N, K in shapes:
        for ring in (2, 3):
            for epi in range(8):
                lib.gemm_set_option("nt_ring", ring)
                got, ins = run_case(M, N, K, epi, 0, seed_t)
                lib.gemm_set_option("nt_ring", 0)
                old, _ = run_case(M, N, K, epi, 0, seed_t)
                ref = reference(M, N, K, epi, ins)
                for k, (g, o) in enumerate(zip(got, old)):
                    scale = float(o.abs().max())
                    d_old = float((g - o).abs().max())
                    d_ref = float((g - ref[k]).abs().max()) if ref is not None else float("nan")
                    ok = d_old <= 1e-6 * max(scale, 1e-6) and bool(torch.isfinite(g).all())      # same arithmetic, same order: bit-equal
                    print("M %6d N %6d K %5d ring %d epi %d out %d: |d two-stage| %.3e |d ref| %.3e scale %.3e %s"
                          % (M, N, K, ring, epi, k, d_old, d_ref, scale, "ok" if ok else "FAIL"), flush=True)
                    bad += 0 if ok else 1
    lib.gemm_set_option("nt_ring", 1)
    print("RING CHECK %s (%d failures)" % ("PASSED" if bad == 0 else "FAILED", bad), flush=True)
    return bad


def bench_ring(batch):
    M = batch * 101
    BT = batch * 64
    shapes = [("qkv fwd", M, 2304, 768, "bias"), ("attn-out fwd", M, 768, 768, "dropres"), ("ffn1 fwd", M, 3072, 768, "gelu"),
              ("ffn2 fwd", M, 768, 3072, "dropres"), ("out dgrad", M, 768, 768, "plain"), ("qkv dgrad", M, 768, 2304, "res"),
              ("ffn1 dgrad", M, 768, 3072, "res"), ("ffn2 dgrad", M, 3072, 768, "mulaux"), ("decoder fwd", BT, 30522, 768, "bias")]
    variants = [("two-stage", dict(p8_mode=0, nt_ring=0)), ("ring 128x128", dict(p8_mode=0, nt_ring=2)), ("ring 128x64", dict(p8_mode=0, nt_ring=3)),
                ("p8 model", dict(p8_mode=1, p8_keepb=1, p8_group=2, p8_min_tiles=160, nt_ring=0))]
    print("%-14s %7s %6s %6s | " % ("gemm", "M", "N", "K") + " | ".join("%-13s" % v[0] for v in variants))
    tot = [0.0] * len(variants)
    best = 0.0
    for name, m, n, k, kwm in shapes:
        row, ts = [], []
        for vi, (vn, opt) in enumerate(variants):
            ms, tf = bench_one(m, n, k, kwm, opt)
            row.append("%6.1fus %5.0f" % (ms * 1e3, tf))
            tot[vi] += ms * (1 if name.startswith("decoder") else 12)
            ts.append(ms * (1 if name.startswith("decoder") else 12))
        best += min(ts)
        print("%-14s %7d %6d %6d | " % (name, m, n, k) + " | ".join(row), flush=True)
    print("%-36s | " % "NT ms per step (12 layers + decoder fwd)" + " | ".join("%13.2f" % t for t in tot) + " | best-of %.2f" % best, flush=True)
    lib.gemm_set_option("nt_ring", 1)


def stagger(batch):
    """128x128 kernel (2 workgroups per CU): phase offset between the two residents of a CU"""
    M = batch * 101
    shapes = [("qkv fwd", M, 2304, 768, "bias"), ("attn-out fwd", M, 768, 768, "dropres"), ("ffn1 fwd", M, 3072, 768, "gelu"),
              ("out dgrad", M, 768, 768, "plain"), ("ffn2 dgrad", M, 3072, 768, "mulaux"), ("ffn2 fwd", M, 768, 3072, "dropres"),
              ("square 4096", 4096, 4096, 4096, "plain")]
    sts = (0, 1, 2, 3, 4, 6, 8)
    print("%-14s | " % "gemm" + " | ".join("stagger %-6d" % s for s in sts))
    for name, m, n, k, kwm in shapes:
        row = []
        for st in sts:
            ms, tf = bench_one(m, n, k, kwm, dict(p8_mode=0, nt_stagger=st))
            row.append("%6.1fus %5.0f" % (ms * 1e3, tf))
        print("%-14s | " % name + " | ".join(row), flush=True)
    lib.gemm_set_option("nt_stagger", 0)


def ablate(batch):
    """where the time of a short-K GEMM goes: full kernel | no epilogue | epilogue without global stores"""
    M = batch * 101
    shapes = [("qkv fwd", M, 2304, 768, "bias"), ("ffn1 fwd", M, 3072, 768, "gelu"), ("ffn2 dgrad", M, 3072, 768, "mulaux"),
              ("attn-out fwd", M, 768, 768, "dropres"), ("ffn2 fwd", M, 768, 3072, "dropres"), ("square 8192", 8192, 8192, 8192, "plain")]
    print("%-14s | %-30s | %-30s" % ("gemm", "tile 256: full / no-epi / no-store (us)", "tile 320: full / no-epi / no-store (us)"))
    for name, m, n, k, kwm in shapes:
        row = []
        for mode in (4, 5):
            t = []
            for ab in (0, 1, 2):
                ms, tf = bench_one(m, n, k, kwm, dict(p8_mode=mode, p8_ablate=ab))
                t.append(ms * 1e3)
            row.append("%7.1f / %7.1f / %7.1f" % tuple(t))
        lib.gemm_set_option("p8_ablate", 0)
        print("%-14s | %-30s | %-30s" % (name, row[0], row[1]), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "ringcheck":
        sys.exit(1 if check_ring() else 0)
    if len(sys.argv) > 1 and sys.argv[1] == "ring":
        bench_ring(int(sys.argv[2]) if len(sys.argv) > 2 else 256)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "stagger":
        stagger(int(sys.argv[2]) if len(sys.argv) > 2 else 256)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "ablate":
        ablate(int(sys.argv[2]) if len(sys.argv) > 2 else 256)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "model":
        bench_model(int(sys.argv[2]) if len(sys.argv) > 2 else 256)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "check":
        sys.exit(1 if check() else 0)
    if len(sys.argv) > 1 and sys.argv[1] == "wgradvision":
        bench_wgrad_vision()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "wgradgroup":
        for b in [int(x) for x in sys.argv[2:]] or [256]:
            bench_wgrad_group(b)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "wgrad":
        bench_wgrad(int(sys.argv[2]) if len(sys.argv) > 2 else 256)
        sys.exit(0)
    bench(int(sys.argv[2]) if len(sys.argv) > 2 else 256)
