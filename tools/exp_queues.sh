set -e
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for q in 4 8; do
    for o in "--s32 la" "--s32 own"; do
      GPU_MAX_HW_QUEUES=$q timeout -k 10 200 python bench.py --cpu-pairs 2 --ref-pairs 0 --no-other-paths --no-clock-probe $o | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('queues $q [$o]', round(d['value'],1), round(d['ms_per_step'],4))"
    done
  done
done
