set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/exp_sq
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/exp_sq -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --pairs 2 --cpu-pairs 0 --no-other-paths --no-clock-probe $EXP_OPTS > $OUT/exp_sq.log 2>&1
python3 - $(find $OUT/exp_sq -name '*counter_collection.csv' | head -1) <<'PY'
import csv, sys, collections, re
cnt = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"^void |x265amd::", "", r["Kernel_Name"])[:40]
    cnt[(k, int(r["Grid_Size"]))][r["Counter_Name"]] += float(r["Counter_Value"])
tot = 0
for (k, g), v in sorted(cnt.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0))[:22]:
    w = v["SQ_WAVES"]
    tot += v["SQ_INSTS_VALU"]
    print("%-42s grid %9d waves %7d  VALU %10.0f (%7.1f/wave) SALU %9.0f LDS %9.0f wavecyc/wave %8.0f wait/wave %8.0f" % (
        k, g, w, v["SQ_INSTS_VALU"], v["SQ_INSTS_VALU"] / w, v["SQ_INSTS_SALU"], v["SQ_INSTS_LDS"], v["SQ_WAVE_CYCLES"] / w, v["SQ_WAIT_INST_ANY"] / w))
print("total VALU of the listed kernels (2 pairs):", tot, "per pair", tot / 2)
PY
