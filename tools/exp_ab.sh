# A/B of bench.py under environment switches: EXP_A / EXP_B are "VAR=value ..." strings (empty = defaults); three runs each
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rep in 1 2 3; do
  for v in A B ${EXP_MORE}; do
    eval "envs=\$EXP_$v"
    r=$(env $envs timeout -k 10 200 python bench.py --cpu-pairs 0 --no-other-paths --no-clock-probe $EXP_OPTS 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f Gpx/s %.3f ms' % (d['value'], d['ms_per_step']))")
    echo "$v [$envs] $r"
  done
done
