"""GEMM launches of a rocprofv3 --kernel-trace run grouped by (kind, grid): python tools/gemm_shapes_from_trace.py <rocpd .db file>"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
rows = cur.execute("select name, grid_x, grid_y, workgroup_x, (end-start) from kernels where name like '%gemm_%'").fetchall()
agg = {}
for n, gx, gy, wx, d in rows:
    k = ('tn' if 'gemm_tn' in n else 'nt', gx // wx, gy)
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += d
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print(k, 'calls/step %.1f' % (a[0] / 5), 'avg_us %.1f' % (a[1] / a[0] / 1e3), 'ms/step %.3f' % (a[1] / 5e6))
