# kernel durations of the exhaustive search per PU size (rocprofv3 kernel trace of tools/esa_bench.py)
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/esa_trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/esa_trace -- python3 $GRAFT_REPO_ROOT/tools/esa_bench.py --reps 2 $EXP_OPTS > $OUT/esa_trace.log 2>&1
python3 - $(find $OUT/esa_trace -name '*kernel_trace.csv' | head -1) <<'PY'
import csv, sys, collections, re
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"^void |x265amd::", "", r["Kernel_Name"])[:48]
    agg[(k, int(r["Grid_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (k, g), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) > 50:
        print("%-50s grid %9d n=%d avg %9.1f us" % (k, g, len(v), sum(v) / len(v)))
PY
