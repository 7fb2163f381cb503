#!/usr/bin/env python
"""Round 6: where does the time of the grouped weight-gradient launch (gemm_tn8_kernel) go?  Runs ON THE GPU BOX.

  python tools/tn8_probe.py [batch] [--ablate]

* correctness of both matrix-instruction forms (option "tn8_m32" 0 / 1) against a torch fp32 product on the encoder's four shapes
  (column sums included) and on an edge shape;
* the layer group (QKV, attention output, FFN1, FFN2 over R = batch * 101 rows padded to 128) timed INTERLEAVED for both forms, with
  the per-workgroup clock stamps (vlb_tn8_set_stamps): sustained shader clock, cycles per K tile, spread over the workgroups;
* --ablate (needs a library built with -DVLB_TN8_PROBE, e.g. VLB_LIB_PATH=vl-bert_amd/csrc/ab/libvlbert_hip.so): the same launch with
  bit 0 no LDS-DMA | bit 1 no fragment reads | bit 2 no MFMAs, in shader CYCLES per K tile (the clock moves with the ablation)."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("vl-bert_amd.ops")
lib = importlib.import_module("vl-bert_amd._lib")
d = "cuda:0"
args = [a for a in sys.argv[1:] if not a.startswith("--")]
B = int(args[0]) if args else 256
R = (B * 101 + 127) // 128 * 128
ABLATE = "--ablate" in sys.argv


def member(Mo, No, rows=R):
    dy = (torch.rand((rows, Mo), device=d) * 2 - 1).to(torch.bfloat16)
    x = (torch.rand((rows, No), device=d) * 2 - 1).to(torch.bfloat16)
    return dy, x, torch.zeros((Mo, No), device=d), torch.zeros(Mo, device=d)


shapes = [("qkv", 2304, 768), ("out", 768, 768), ("ffn1", 3072, 768), ("ffn2", 768, 3072)]
M = [member(mo, no) for _, mo, no in shapes]
ws = torch.empty(3 * sum(mo * no for _, mo, no in shapes) + 1024, device=d)
flops = 2.0 * R * sum(mo * no for _, mo, no in shapes)
stamps = torch.zeros(256 * 4, dtype=torch.int64, device=d)


def launch():
    ops.wgrad_tn_group(M, workspace=ws, accumulate=False)


VARIANTS = [(0, 0), (1, 0)]      # (tn8_m32, reserved)


def select(m32, sched):
    lib.gemm_set_option("tn8_m32", m32)


def check():
    ok = True
    for m32, sched in VARIANTS:
        select(m32, sched)
        for t in M:
            t[2].zero_(); t[3].zero_()
        launch()
        torch.cuda.synchronize()
        for (name, mo, no), (dy, x, C, cs) in zip(shapes, M):
            ref = dy.float().t() @ x.float()
            err = (C - ref).abs().max().item() / ref.abs().max().item()
            cerr = (cs - dy.float().sum(0)).abs().max().item() / dy.float().sum(0).abs().max().item()
            good = err < 2e-5 and cerr < 1e-4
            ok &= good
            print("check m32=%d sched=%d %-5s rel err %.2e colsum %.2e %s" % (m32, sched, name, err, cerr, "ok" if good else "BAD"))
        # edge shape: Mo, No not multiples of 256 / 16, accumulate into a non-zero C, through the single-gradient entry
        dy, x, C, cs = member(1000, 520, 4096)
        C.fill_(0.5)
        ops.wgrad_tn_group([(dy, x, C, cs)], workspace=ws, accumulate=True)
        torch.cuda.synchronize()
        ref = dy.float().t() @ x.float() + 0.5
        err = (C - ref).abs().max().item() / ref.abs().max().item()
        cerr = (cs - dy.float().sum(0)).abs().max().item()
        good = err < 2e-5 and cerr < 1e-2
        ok &= good
        print("check m32=%d sched=%d edge  rel err %.2e colsum abs %.2e %s" % (m32, sched, err, cerr, "ok" if good else "BAD"))
    return ok


def timed(iters=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        launch()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def stamped():
    """one launch with the per-workgroup stamps: (sustained MHz, median / max cycles per workgroup, workgroups)"""
    stamps.zero_()
    lib.call("vlb_tn8_set_stamps", stamps.data_ptr())
    launch()
    torch.cuda.synchronize()
    lib.call("vlb_tn8_set_stamps", None)
    s = stamps.view(256, 4).cpu()
    s = s[s[:, 2] > 0]
    cyc = (s[:, 2] - s[:, 0]).double()
    ticks = (s[:, 3] - s[:, 1]).double()
    mhz = (cyc / ticks * 100.0).median().item()
    stamped.last = cyc
    return mhz, cyc.median().item(), cyc.max().item(), len(s)


print("tn8 probe: batch %d, R = %d, %.3f TFLOP per launch" % (B, R, flops / 1e12))
if not check():
    print("CHECK FAILED")
for _ in range(10):
    launch()
kt = R / 64 / 2          # K tiles per work item (two K slices)
for rnd in range(4):
    for m32, sched in VARIANTS:
        select(m32, sched)
        us = timed()
        mhz, med, mx, n = stamped()
        print("round %d m32=%d sched=%d: %7.1f us  %7.1f TFLOP/s | %4.0f MHz, %d workgroups, cycles per K tile median %.0f max %.0f"
              % (rnd, m32, sched, us, flops / us / 1e6, mhz, n, med / kt, mx / kt))
# which workgroups are the slow ones?  (block b runs on XCD b % 8; the item list is XCD-contiguous, see the kernel)
cyc = stamped.last / kt
order = torch.argsort(cyc, descending=True)
print("slowest workgroups (block: cycles per K tile):", ", ".join("%d: %.0f" % (int(i), cyc[i].item()) for i in order[:24]))
print("fastest workgroups:", ", ".join("%d: %.0f" % (int(i), cyc[i].item()) for i in order[-12:]))
for x in range(8):
    sel = cyc[x::8]
    print("  XCD %d: median %.0f max %.0f" % (x, sel.median().item(), sel.max().item()))
# TWO layers' weight gradients as ONE table launch of 216 full-K items (no K slices, no slabs, no reduce launch) against two grouped launches
M2 = [member(mo, no) for _, mo, no in shapes]
tab = ops.WgradTable([(a, b, c, cs, None) for a, b, c, cs in M + M2], d, accumulate=False)
assert tab.ok, "table refused"
for t in M + M2:
    t[2].zero_(); t[3].zero_()
tab.run()
torch.cuda.synchronize()
for (name, mo, no), (dy, x, C, cs) in zip(shapes + shapes, M + M2):
    ref = dy.float().t() @ x.float()
    err = (C - ref).abs().max().item() / ref.abs().max().item()
    print("pair table %-5s rel err %.2e %s" % (name, err, "ok" if err < 2e-5 else "BAD"))
for m32, sched in ((0, 0), (1, 0)):
    select(m32, sched)
    for rnd in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            tab.run()
        e0.record()
        for _ in range(10):
            tab.run()
        e1.record()
        torch.cuda.synchronize()
        us_tab = e0.elapsed_time(e1) / 10 * 1e3
        MM = M
        for _ in range(3):
            launch()
        e0.record()
        for _ in range(10):
            M = MM; launch(); M = M2; launch()
        e1.record()
        torch.cuda.synchronize()
        M = MM
        us_two = e0.elapsed_time(e1) / 10 * 1e3
        print("m32=%d sched=%d two layers: table launch %7.1f us (%6.1f TFLOP/s) | two grouped launches + reduces %7.1f us (%6.1f TFLOP/s)"
              % (m32, sched, us_tab, 2 * flops / us_tab / 1e6, us_two, 2 * flops / us_two / 1e6))
select(0, 0)

# the same launch WITHOUT the bias-gradient column sums: are the n0 = 0 tiles (whose wn = 0 waves take them) the slow workgroups?
M_cs = M
M = [(a, b, c, None) for a, b, c, _ in M_cs]
for rnd in range(2):
    us = timed()
    mhz, med, mx, n = stamped()
    print("no column sums, round %d: %7.1f us | %4.0f MHz, cycles per K tile median %.0f max %.0f" % (rnd, us, mhz, med / kt, mx / kt))
cyc = stamped.last / kt
order = torch.argsort(cyc, descending=True)
print("  slowest:", ", ".join("%d: %.0f" % (int(i), cyc[i].item()) for i in order[:16]))
M = M_cs
if ABLATE:
    names = {0: "full", 1: "no DMA", 2: "no reads", 3: "no DMA, no reads (MFMA + barriers)", 4: "no MFMA", 5: "no MFMA, no DMA (reads + barriers)",
             6: "no MFMA, no reads (DMA + barriers)", 8: "all workgroups stream item 0's panels (L2 hits)", 12: "no MFMA, item 0's panels (reads + DMA + barriers, L2 hits)", 14: "DMA + barriers, item 0's panels"}
    for m32, sched in ((0, 0), (1, 0)):
        select(m32, sched)
        for ab in (0, 1, 2, 3, 4, 5, 6, 8, 12, 14):
            lib.gemm_set_option("tn8_ablate", ab)
            for _ in range(3):
                launch()
            us = timed(10)
            mhz, med, mx, n = stamped()
            print("ablate m32=%d sched=%d %-40s: %7.1f us | %4.0f MHz, cycles per K tile median %.0f max %.0f"
                  % (m32, sched, names[ab], us, mhz, med / kt, mx / kt))
    lib.gemm_set_option("tn8_ablate", 0)
select(0, 0)
