set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest_final.log 2>&1
timeout -k 10 400 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_final -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/prof_final.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --pairs 2 --cpu-pairs 0 --no-other-paths > $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch2.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_write2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --pairs 2 --cpu-pairs 0 --no-other-paths > $GRAFT_REPO_ROOT/gpurun_out/pmc_write2.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_sq2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --pairs 2 --cpu-pairs 0 --no-other-paths > $GRAFT_REPO_ROOT/gpurun_out/pmc_sq2.log 2>&1
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python -u tools/mc_bench.py > gpurun_out/mc_bench_final.txt 2>&1
timeout -k 10 200 python -u tools/mc_bench.py --width 3840 --height 2160 --no-check >> gpurun_out/mc_bench_final.txt 2>&1
timeout -k 10 200 python -u tools/mc_bench.py --width 3840 --height 2160 --depth 10 --no-check >> gpurun_out/mc_bench_final.txt 2>&1
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_mc -- python3 $GRAFT_REPO_ROOT/tools/mc_bench.py --width 3840 --height 2160 --no-check > $GRAFT_REPO_ROOT/gpurun_out/prof_mc.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_bipred_final -- python3 $GRAFT_REPO_ROOT/tools/bipred_bench.py > $GRAFT_REPO_ROOT/gpurun_out/bipred_bench_final.txt 2>&1
