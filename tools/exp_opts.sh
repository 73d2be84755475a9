# bench.py under several option sets: EXP_SETS = "opts1|opts2|..." (each a bench.py option string); EXP_REPS rounds
cd "${GRAFT_REPO_ROOT:-/root/repo}"
IFS='|' read -ra SETS <<< "$EXP_SETS"
for rep in $(seq 1 ${EXP_REPS:-2}); do
  for o in "${SETS[@]}"; do
    r=$(timeout -k 10 200 python bench.py --cpu-pairs 0 --no-other-paths --no-clock-probe $o 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f Gpx/s %.3f ms' % (d['value'], d['ms_per_step']))")
    echo "[$o] $r"
  done
done
