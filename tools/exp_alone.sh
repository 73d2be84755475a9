# kernel durations of launch classes of the headline step running alone on the chip (EXP_ONLY = 16 | 64 | la | 16,64 ...)
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for only in ${EXP_ONLY:-16 64 la}; do
  rm -rf $OUT/exp_alone
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/exp_alone -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --cpu-pairs 0 --no-other-paths --no-clock-probe --only $only > $OUT/exp_alone.log 2>&1
  echo "== only $only"
  python3 $GRAFT_REPO_ROOT/tools/step_timeline.py $(find $OUT/exp_alone -name '*kernel_trace.csv' | head -1) || true
done
