# The rocprofv3 passes behind the roofline numbers of bench.py, one run per pass (run on the GPU box from the repo root):
#   bash tools/profile_round.sh <tag>      -> gpurun_out/<tag>_{trace,sq,fetch,write}/...
# Summaries for profiles/ are made afterwards with tools/profile_summary.py.
set -e
TAG=${1:-r03}
OUT=$GRAFT_REPO_ROOT/gpurun_out
B=$GRAFT_REPO_ROOT/bench.py
cd /tmp && export TMPDIR=/tmp
# the counter passes run a 2-pair step: its lookahead launch would take the form for small launches; keep the form of the 32-pair step
export X265AMD_LOOKAHEAD_ROWS=4
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -- python3 $B --steps 5 --warmup 2 --cpu-pairs 0 --no-other-paths --no-clock-probe > $OUT/${TAG}_trace.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/${TAG}_sq -- python3 $B --steps 1 --warmup 0 --pairs 2 --cpu-pairs 0 --no-other-paths --no-clock-probe > $OUT/${TAG}_sq.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_fetch -- python3 $B --steps 1 --warmup 0 --pairs 2 --cpu-pairs 0 --no-other-paths --no-clock-probe > $OUT/${TAG}_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_write -- python3 $B --steps 1 --warmup 0 --pairs 2 --cpu-pairs 0 --no-other-paths --no-clock-probe > $OUT/${TAG}_write.log 2>&1
