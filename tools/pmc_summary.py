#!/usr/bin/env python
"""Per-dispatch-shape PMC sums for kernels matching a pattern (rocprofv3 rocpd sqlite)."""
import sqlite3, sys
db, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "gemm_nt")
con = sqlite3.connect(db)
cur = con.cursor()
rows = cur.execute("select dispatch_id, grid_size_x, grid_size_y, workgroup_size_x, counter_name, value, duration from counters_collection "
                   "where kernel_name like ?", ("%" + pat + "%",)).fetchall()
disp = {}
for did, gx, gy, wx, name, val, dur in rows:
    d = disp.setdefault(did, {"shape": (gx // wx, gy), "dur": dur, "c": {}})
    d["c"][name] = d["c"].get(name, 0.0) + val
agg = {}
for d in disp.values():
    a = agg.setdefault(d["shape"], {"n": 0, "dur": 0.0, "c": {}})
    a["n"] += 1
    a["dur"] += d["dur"]
    for k, v in d["c"].items():
        a["c"][k] = a["c"].get(k, 0.0) + v
names = sorted({k for a in agg.values() for k in a["c"]})
print("shape(blocks,splits)  n  avg_us  " + "  ".join(names))
for shape, a in sorted(agg.items(), key=lambda kv: -kv[1]["dur"]):
    print(shape, a["n"], "%.1f" % (a["dur"] / a["n"] / 1e3), "  ".join("%.4g" % (a["c"].get(k, 0) / a["n"]) for k in names))
