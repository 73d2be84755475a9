"""Per-kernel summary of a rocprofv3 --kernel-trace run stored as a rocpd SQLite database:
calls, total / average / min / max duration in microseconds, grid and workgroup size, LDS, VGPRs.

    python profiles/db_summary.py gpurun_out/prof_xyz/<host>/<pid>_results.db
"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.execute("select * from kernels")
cols = [d[0] for d in cur.description]
agg = collections.defaultdict(list)
for row in cur:
    r = dict(zip(cols, row))
    key = (r["name"][:60], r.get("grid_x"), r.get("workgroup_x"), r.get("lds_size"), r.get("vgpr_count", r.get("arch_vgpr_count")))
    agg[key].append((r["end"] - r["start"]) / 1e3)
print("%-60s %8s %6s %7s %5s %6s %10s %9s %9s %9s" % ("kernel", "grid", "wg", "lds", "vgpr", "calls", "total_us", "avg_us", "min_us", "max_us"))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print("%-60s %8s %6s %7s %5s %6d %10.1f %9.1f %9.1f %9.1f" % (k + (len(v), sum(v), sum(v) / len(v), min(v), max(v))))
