set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/exp_trace
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/exp_trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --cpu-pairs 0 --no-other-paths --no-clock-probe $EXP_OPTS > $OUT/exp_trace.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/step_timeline.py $(find $OUT/exp_trace -name '*kernel_trace.csv' | head -1)
