#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite) kernel trace: per-kernel calls / total / avg / share.
Usage: rocprof_summary.py results.db [steps]   (divide totals by `steps` to get per-step numbers)"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:70]


def main():
    db = sys.argv[1]
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else "kernel_name"
    rows = cur.execute("select %s, (end - start) as dur, grid_x*1.0, workgroup_x from kernels" % namecol
                       if "grid_x" in cols else "select %s, (end - start) as dur, 0, 0 from kernels" % namecol).fetchall()
    agg = {}
    for name, dur, gx, wx in rows:
        k = short(name)
        a = agg.setdefault(k, [0, 0.0, 1e30, 0.0])
        a[0] += 1
        a[1] += dur
        a[2] = min(a[2], dur)
        a[3] = max(a[3], dur)
    total = sum(a[1] for a in agg.values())
    print("%-72s %8s %12s %10s %10s %10s %7s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "share"))
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-72s %8d %12.3f %10.2f %10.2f %10.2f %6.1f%%" % (k, a[0], a[1] / 1e6, a[1] / a[0] / 1e3, a[2] / 1e3, a[3] / 1e3,
                                                                 100 * a[1] / total))
    print("TOTAL kernel time %.3f ms over %g steps -> %.3f ms/step" % (total / 1e6, steps, total / 1e6 / steps))


if __name__ == "__main__":
    main()
