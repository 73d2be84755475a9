# A/B of engine builds on the headline step: the in-tree library against the libraries named in $EXP_LIBS, alternating, four runs each
set -e
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4; do
  for lib in "" $EXP_LIBS; do
    X265AMD_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} timeout -k 10 200 python bench.py --cpu-pairs 2 --ref-pairs 0 --no-other-paths --no-clock-probe | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('${lib:-in-tree}', round(d['value'],1), round(d['ms_per_step'],4))"
  done
done
