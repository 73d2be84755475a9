#!/usr/bin/env python3
"""Sum of every counter per kernel of a rocprofv3 --pmc csv run: python tools/pmc_sum.py <dir> [kernel substring]"""
import csv, glob, sys, collections
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else ""
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        if sub not in k: continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
for k in tot:
    print(k)
    for c, v in sorted(tot[k].items()): print("   %-34s %16.0f  (%d launches)" % (c, v, n[(k, c)]))
