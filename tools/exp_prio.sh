set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests/test_gpu_search.py tests/test_gpu_configs.py -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3
for opt in "--s32 la" "--s32 big" "--s32 mid" "--s32 own" "--s32 la"; do
  echo "== $opt"
  timeout -k 10 200 python bench.py --cpu-pairs 2 --ref-pairs 0 --no-other-paths $opt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'])"
done
EXP_OPTS="--s32 la" bash tools/exp_trace.sh
bash tools/exp_sq.sh
