"""Per-launch kernel durations of tools/bipred_bench.py under rocprofv3 --kernel-trace:
python profiles/bipred_trace_summary.py <kernel_trace.csv> > profiles/r01_bipred_bench.txt"""
import collections
import csv
import sys

print("tools/bipred_bench.py under rocprofv3 --kernel-trace (1920x1080 8-bit, every PU / CU of one picture per launch, random")
print("quarter-pel MVs; merge candidates: one third each list 0, list 1, both). Kernel durations per launch from the trace.")
print("bipred_kernel = one wave per PU (device-resident jobs); bipred_flat_kernel, merge_flat_kernel = 64 tiles per wave across")
print("PUs (host job arrays).\n")
d = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    if "bipred" in r["Kernel_Name"] or "merge" in r["Kernel_Name"]:
        name = r["Kernel_Name"].split("<")[0].replace("void x265amd::", "")
        d.setdefault((name, int(r["Grid_Size_X"])), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
order = ["8x8, 32400", "16x16, 8040", "32x32, 1980", "64x64, 480"]
seen = collections.Counter()
for (k, g), v in d.items():
    label = order[min(seen[k], 3)]
    seen[k] += 1
    print("%-20s %-12s per launch (grid %8d threads): n=%d median %.1f us min %.1f us" % (k, label, g, len(v), sorted(v)[len(v) // 2], min(v)))
