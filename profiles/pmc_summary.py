"""Per-kernel totals of a rocprofv3 --pmc run: python profiles/pmc_summary.py <pmc_counter_collection.csv> [name-substring]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
want = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    if want in r["Kernel_Name"]:
        agg[(r["Kernel_Name"][:48], r["Grid_Size"])][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in agg.items():
    waves = v.get("SQ_WAVES", 0)
    print(k)
    for c, x in sorted(v.items()):
        print("   %-22s %12.4g   per wave %10.1f" % (c, x, x / waves if waves else float("nan")))
