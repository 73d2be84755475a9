# Experiment: SQ counters of the exhaustive-search kernels per PU size (4K, merange 92). Run on the GPU box.
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/esa_pmc -- python3 $R/tools/esa_bench.py --reps 1 > $R/gpurun_out/esa_pmc.log 2>&1
