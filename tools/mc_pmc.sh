set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD --output-format csv -d $R/gpurun_out/pmc_mc_sq -- python3 $R/tools/mc_bench.py --width 3840 --height 2160 --no-check --reps 1 > $R/gpurun_out/pmc_mc_sq.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_mc_fetch -- python3 $R/tools/mc_bench.py --width 3840 --height 2160 --no-check --reps 1 > $R/gpurun_out/pmc_mc_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_mc_write -- python3 $R/tools/mc_bench.py --width 3840 --height 2160 --no-check --reps 1 > $R/gpurun_out/pmc_mc_write.log 2>&1
