# A/B of an environment switch on the headline step: four runs with and without $EXP_ENV (e.g. X265AMD_NO_LEAN=1)
set -e
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4; do
  timeout -k 10 200 python bench.py --cpu-pairs 2 --ref-pairs 0 --no-other-paths --no-clock-probe | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('default', round(d['value'],1), round(d['ms_per_step'],4))"
  env $EXP_ENV timeout -k 10 200 python bench.py --cpu-pairs 2 --ref-pairs 0 --no-other-paths --no-clock-probe | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$EXP_ENV', round(d['value'],1), round(d['ms_per_step'],4))"
done
