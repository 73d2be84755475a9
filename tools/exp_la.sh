# A/B of engine builds on the lookahead paths: the in-tree library against the libraries named in $EXP_LIBS, alternating, three runs each
set -e
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for lib in "" $EXP_LIBS; do
    X265AMD_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} timeout -k 10 200 python bench.py --workload la8k --cpu-pairs 0 --ref-pairs 0 --no-other-paths --no-clock-probe | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('${lib:-in-tree}', 'la8k', round(d['value'],1), round(d['ms_per_step'],4))"
    X265AMD_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} timeout -k 10 200 python tools/la_bench.py --frames 16 --reps 3 2>&1 | grep -v amdgpu.ids | tail -2
  done
done
