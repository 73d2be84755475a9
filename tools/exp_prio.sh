set -e
cd $GRAFT_REPO_ROOT
for opt in "--prio small" "--prio big" "--prio flat" "--prio big --pairs 64" "--prio small --pairs 64"; do
  echo "== $opt"
  timeout -k 10 200 python bench.py --cpu-pairs 0 --no-other-paths $opt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'])"
done
