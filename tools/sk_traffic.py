#!/usr/bin/env python
"""HBM read traffic of the N = 768 long-K GEMM under the three decompositions -- the measurement behind "pure stream-K loses the L2"
(csrc/gemm.hip, header of gemm_nt_sk_kernel).  Runs ON THE GPU BOX:
    python tools/sk_traffic.py <out.txt>              driver: one rocprofv3 --kernel-trace --pmc FETCH_SIZE pass per mode, then the table
    python tools/sk_traffic.py --run <mode> <M>       the traced workload: 20 launches of the FFN1-data-gradient form (A[M,3072] W[768,3072]^T + residual)
FETCH_SIZE is doubled (gfx950 counts a 128-B request of a 16-B/lane streaming read as 64 B -- MI355X_MICROARCH.md, HBM section), as in
tools/profile_report.py.  Algorithmic operand bytes: (M + 768) x 3072 x 2."""
import glob
import importlib
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
MODES = ((0, "whole tiles (ring 128x64)"), (2, "hybrid stream-K"), (3, "pure stream-K"))
SIZES = (3232, 6464)


def run(mode, M):
    import torch
    L = importlib.import_module("vl-bert_amd._lib")
    ops = importlib.import_module("vl-bert_amd.ops")
    g = torch.Generator(device="cuda:0").manual_seed(0)
    A = (torch.randn(M, 3072, device="cuda:0", generator=g) * 0.5).to(ops.BF16)
    W = (torch.randn(768, 3072, device="cuda:0", generator=g) * 0.05).to(ops.BF16)
    res = torch.randn(M, 768, device="cuda:0", generator=g).to(ops.BF16)
    C = torch.empty(M, 768, dtype=ops.BF16, device="cuda:0")
    L.gemm_set_option("nt_sk", mode)
    for _ in range(20):
        ops.gemm_nt(A, W, C, res=res)
    torch.cuda.synchronize()


def main():
    if sys.argv[1] == "--run":
        return run(int(sys.argv[2]), int(sys.argv[3]))
    out = sys.argv[1]
    rows = []
    for M in SIZES:
        for mode, label in MODES:
            d = "/tmp/sk_traffic_%d_%d" % (mode, M)
            subprocess.run(["rm", "-rf", d])
            subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", "FETCH_SIZE", "-d", d, "-o", "r", "--", sys.executable, os.path.abspath(__file__),
                            "--run", str(mode), str(M)], cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL, timeout=600)
            db = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)[0]
            cur = sqlite3.connect(db).cursor()
            per = {}
            for did, kn, v, dur in cur.execute("select dispatch_id, kernel_name, value, duration from counters_collection where counter_name = 'FETCH_SIZE'"):
                if "gemm_nt" in kn:
                    a = per.setdefault(did, [kn, 0.0, dur])
                    a[1] += v
            vals = list(per.values())[2:]                      # (skip the first launches: cold caches)
            kb = sum(v[1] for v in vals) / max(len(vals), 1)
            us = sum(v[2] for v in vals) / max(len(vals), 1) / 1e3
            name = vals[0][0].split("(")[0].replace("void ", "")[:60] if vals else "?"
            rows.append((M, label, name, 2 * kb / 1e3, us))
            subprocess.run(["rm", "-rf", d])
    with open(out, "w") as f:
        f.write("# HBM read traffic per launch (rocprofv3 --pmc FETCH_SIZE, x2: gfx950 streaming-read correction), C[M,768] = A[M,3072] W[768,3072]^T + residual,\n"
                "# 18 launches after 2 cold ones, per decomposition (option nt_sk: 0 / 2 / 3); algorithmic operand bytes = (M + 768) x 3072 x 2\n")
        f.write("%6s  %-28s %-62s %12s %14s %9s\n" % ("M", "decomposition", "kernel", "read MB", "algorithmic MB", "us (traced)"))
        for M, label, name, mb, us in rows:
            f.write("%6d  %-28s %-62s %12.1f %14.1f %9.1f\n" % (M, label, name, mb, (M + 768) * 3072 * 2 / 1e6, us))
    print(open(out).read())


if __name__ == "__main__":
    main()
