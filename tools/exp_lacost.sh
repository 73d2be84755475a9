# A/B of engine builds on the lookahead frame-cost bench (1080p and 8K): in-tree against $EXP_LIBS, three runs each
set -e
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for lib in "" $EXP_LIBS; do
    echo "== ${lib:-in-tree}"
    X265AMD_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} timeout -k 10 200 python tools/la_bench.py --frames 16 --reps 3 2>&1 | grep -v amdgpu.ids | tail -1
    X265AMD_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} timeout -k 10 200 python tools/la_bench.py --width 7680 --height 4320 --frames 4 --reps 3 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
