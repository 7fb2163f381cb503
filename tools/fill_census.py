#!/usr/bin/env python
"""How many torch fill / copy / random launches does one STEADY-STATE step issue?  (Reviews of rounds 2-4: "2062 FillFunctor launches
over the 5-step process -- separate one-time set-up from per-step work.")

Runs ON THE GPU BOX: traces the same bench command twice with rocprofv3 --kernel-trace, with K and K + D timed steps, and reports
(launches in run 2 - launches in run 1) / D per kernel family: set-up cancels, what remains is issued every step.
    python tools/fill_census.py <out.txt> [--e2e]"""
import glob
import os
import re
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_path = sys.argv[1]
e2e = "--e2e" in sys.argv
K, D = 2, 6


def trace(steps, tag):
    d = "/tmp/fill_census_%s" % tag
    subprocess.run(["rm", "-rf", d])
    cmd = ["rocprofv3", "--kernel-trace", "-d", d, "-o", "r", "--", sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps),
           "--warmup", "1", "--no-graph", "--no-cpu-baseline", "--no-phase-times", "--no-clock-probe", "--head-start", "0"] + (["--e2e"] if e2e else [])
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
    db = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)[0]
    agg = {}
    for n, dur in sqlite3.connect(db).cursor().execute("select name, end - start from kernels").fetchall():
        n = re.sub(r"^void ", "", re.sub(r"\(.*$", "", n.replace("(anonymous namespace)::", "")))[:90]
        a = agg.setdefault(n, [0, 0.0])
        a[0] += 1
        a[1] += dur
    subprocess.run(["rm", "-rf", d])
    return agg


a, b = trace(K, "a"), trace(K + D, "b")
rows = []
for n in sorted(set(a) | set(b)):
    ca, ta = a.get(n, [0, 0.0])
    cb, tb = b.get(n, [0, 0.0])
    rows.append((n, (cb - ca) / D, (tb - ta) / D / 1e3, ca - K * (cb - ca) / D))
torchish = [r for r in rows if r[0].startswith(("at::", "__amd_rocclr", "Cijk"))]
with open(out_path, "w") as f:
    f.write("# %s -- launches per STEADY-STATE step = (launches with %d timed steps - launches with %d) / %d; rocprofv3 --kernel-trace of\n"
            "# python bench.py --steps N --warmup 1 --no-graph --no-cpu-baseline --no-phase-times --no-clock-probe --head-start 0%s\n"
            "# 'set-up' = launches of the shorter run that the per-step rate does not explain (allocation fills, random initialisation, warm-up extras)\n"
            % ("e2e configuration (C3)" if e2e else "headline workload (global batch 256)", K + D, K, D, " --e2e" if e2e else ""))
    f.write("%-92s %12s %12s %10s\n" % ("kernel", "per step", "us per step", "set-up"))
    for n, per, us, setup in sorted(torchish, key=lambda r: -r[1]):
        f.write("%-92s %12.2f %12.1f %10.0f\n" % (n, per, us, setup))
    own = [r for r in rows if r not in torchish]
    f.write("%-92s %12.2f %12.1f\n" % ("(all library kernels together)", sum(r[1] for r in own), sum(r[2] for r in own)))
    f.write("%-92s %12.2f %12.1f\n" % ("(all torch / runtime kernels together)", sum(r[1] for r in torchish), sum(r[2] for r in torchish)))
print(open(out_path).read())
