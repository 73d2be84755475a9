# Experiment: VALU / SALU / LDS instructions per PU of the pattern search; with stage-stop builds under
# build_exp/ (lib_stop<n>.so, -DX265AMD_STAGE_STOP=n) also after each stage. Run on the GPU box.
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for n in ${STAGES:-0}; do
  lib=""; [ $n != 0 ] && lib="--lib $R/build_exp/lib_stop$n.so"
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $R/gpurun_out/stg_pmc$n -- python3 $R/tools/search_bench.py --reps 1 $lib > $R/gpurun_out/stg_pmc$n.log 2>&1
done
