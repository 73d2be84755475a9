# bench.py under alternative builds of the library: EXP_LIBS = names of alt/<name>.so (built beforehand); EXP_REPS rounds
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rep in $(seq 1 ${EXP_REPS:-2}); do
  for v in ${EXP_LIBS}; do
    r=$(X265AMD_LIB=$PWD/alt/$v.so timeout -k 10 200 python bench.py --cpu-pairs 0 --no-other-paths --no-clock-probe $EXP_OPTS 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f Gpx/s %.3f ms' % (d['value'], d['ms_per_step']))")
    echo "$v $r"
  done
done
