# The lookahead band kernel with four, two and one CU row per wave (X265AMD_LOOKAHEAD_ROWS): parity, then every bench
# the lookahead takes part in. $EXP_LIB names the library (default: in-tree).
set -e
cd $GRAFT_REPO_ROOT
export X265AMD_LIB=${EXP_LIB:+$GRAFT_REPO_ROOT/$EXP_LIB}
for r in 4 2 1; do
  export X265AMD_LOOKAHEAD_ROWS=$r
  echo "=== rows per wave: $r"
  timeout -k 10 300 python -u -m pytest tests/test_gpu_lookahead.py tests/test_gpu_configs.py -x -q -m gpu --timeout 120 --timeout-method thread 2>&1 | tail -1
  timeout -k 10 300 python -u tools/soak_lookahead.py --fields 600 --seed $((30 + r)) 2>&1 | tail -1
  for i in 1 2; do
    timeout -k 10 200 python tools/la_bench.py --frames 16 --reps 3 2>&1 | grep -v amdgpu.ids | tail -1
    timeout -k 10 200 python tools/la_bench.py --width 7680 --height 4320 --frames 4 --reps 3 2>&1 | grep -v amdgpu.ids | tail -1
    timeout -k 10 200 python bench.py --workload la8k --cpu-pairs 0 --ref-pairs 0 --no-other-paths --no-clock-probe 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('la8k', round(d['ms_per_step'],3))"
    timeout -k 10 200 python bench.py --cpu-pairs 2 --ref-pairs 0 --no-other-paths --no-clock-probe 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('headline', round(d['value'],1), round(d['ms_per_step'],4))"
  done
done
