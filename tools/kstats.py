#!/usr/bin/env python
"""Per-kernel time of one rocprofv3 --kernel-trace run (rocpd sqlite):  python tools/kstats.py <output dir> <steps in run> [top]"""
import glob
import os
import re
import sqlite3
import sys

src, steps = sys.argv[1], float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
path = glob.glob(os.path.join(src, "**", "*.db"), recursive=True)[0]
cur = sqlite3.connect(path).cursor()
agg = {}
for n, d in cur.execute("select name, end - start from kernels").fetchall():
    n = re.sub(r"^void ", "", re.sub(r"\(.*$", "", n.replace("(anonymous namespace)::", "")))[:72]
    a = agg.setdefault(n, [0, 0.0])
    a[0] += 1; a[1] += d
tot = sum(a[1] for a in agg.values())
print("all kernels: %.3f ms/step over %d launches/step" % (tot / 1e6 / steps, sum(a[0] for a in agg.values()) / steps))
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print("%-72s %7.1f/step %8.3f ms/step %9.2f us avg %5.1f%%" % (n, a[0] / steps, a[1] / 1e6 / steps, a[1] / a[0] / 1e3, 100 * a[1] / tot))
