#!/usr/bin/env python3
"""Timeline of the last step of a traced bench.py run: every kernel between the starts of the last two lookahead launches,
with start / end relative to the step's first kernel.   usage: step_timeline.py kernel_trace.csv"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").replace("x265amd::", "")[:44],
             int(r["Grid_Size_X"])) for r in rows)
las = [e for e in ev if "lookahead_band" in e[2]]
if len(las) < 2:
    las = [e for e in ev if "star_small_kernel<8" in e[2]]      # a run without the lookahead launch
if len(las) < 2:
    las = [e for e in ev if "star_small_kernel<16" in e[2]]
if len(las) < 2:
    las = [e for e in ev if "star_big_kernel<64" in e[2]]
if len(las) < 2:
    las = [e for e in ev if "search_kernel" in e[2] and e[3] > 1000000]
t0, prev = las[-1][0], las[-2][0]
print("step period %.3f ms" % ((t0 - prev) / 1e6))
for e in ev:
    if prev <= e[0] < t0:
        print("%8.3f -> %8.3f  (%7.3f ms) %-44s %d" % ((e[0] - prev) / 1e6, (e[1] - prev) / 1e6, (e[1] - e[0]) / 1e6, e[2], e[3]))
