set -e
cd $GRAFT_REPO_ROOT
for opt in "--s32 big" "--s32 mid" "--s32 la" "--s32 own" "--s32 mid --prio flat" "--s32 own --prio flat"; do
  echo "== $opt"
  timeout -k 10 200 python bench.py --cpu-pairs 2 --ref-pairs 0 --no-other-paths $opt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'])"
done
