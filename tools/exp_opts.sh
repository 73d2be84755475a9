# bench.py option variants on the headline step, alternating, three runs each: EXP_VARIANTS="opts1|opts2|..."
set -e
cd $GRAFT_REPO_ROOT
IFS='|' read -ra V <<< "$EXP_VARIANTS"
for i in 1 2 3; do
  for o in "${V[@]}"; do
    timeout -k 10 200 python bench.py --cpu-pairs 2 --ref-pairs 0 --no-other-paths --no-clock-probe $o | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$o]', round(d['value'],1), round(d['ms_per_step'],4))"
  done
done
