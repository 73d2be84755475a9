# parity of the search paths, then the headline step, then its SQ counters (experiments; EXP_OPTS = bench.py flags)
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests/test_gpu_search.py tests/test_gpu_configs.py -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3
timeout -k 10 200 python bench.py --cpu-pairs 4 --ref-pairs 0 --no-other-paths $EXP_OPTS | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('BENCH', d['value'], d['ms_per_step'], d['roofline']['launch_ms'])"
bash tools/exp_sq.sh
