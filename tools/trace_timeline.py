#!/usr/bin/env python3
"""Per-queue timeline of a rocprofv3 --kernel-trace CSV (kernel_trace.csv): for every queue the busy time, the span from
its first start to its last end, and the kernels by total time; then how many kernels ran concurrently over the trace.
usage: trace_timeline.py kernel_trace.csv [t0_us t1_us]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"][:34],
       int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))) for r in rows]
t0 = min(e[0] for e in ev)
if len(sys.argv) > 3:
    lo, hi = t0 + int(float(sys.argv[2]) * 1e3), t0 + int(float(sys.argv[3]) * 1e3)
    ev = [e for e in ev if e[0] >= lo and e[1] <= hi]
by_q = collections.defaultdict(list)
for e in ev:
    by_q[e[2]].append(e)
print("trace span %.3f ms, %d kernels" % ((max(e[1] for e in ev) - min(e[0] for e in ev)) / 1e6, len(ev)))
for q, es in sorted(by_q.items()):
    es.sort()
    busy = sum(e[1] - e[0] for e in es)
    print("queue %s: %d kernels, first start %.3f ms, last end %.3f ms, busy %.3f ms" %
          (q, len(es), (es[0][0] - t0) / 1e6, (max(e[1] for e in es) - t0) / 1e6, busy / 1e6))
    agg = collections.defaultdict(list)
    for e in es:
        agg[(e[3], e[4])].append((e[1] - e[0]) / 1e3)
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print("    %-34s grid %-7d n=%-4d avg %8.1f us  total %8.3f ms" % (k[0], k[1], len(v), sum(v) / len(v), sum(v) / 1e3))
