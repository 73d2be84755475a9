# Everything measured for a round's profiles/ in one go (run on the GPU box from the repo root):
#   bash tools/round_measure.sh r02
# then, back in the container: python tools/profile_summary.py gpurun_out r02, and copy the text outputs to profiles/.
set -e
TAG=${1:-r02}
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/${TAG}_pytest.log 2>&1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1
timeout -k 10 400 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
timeout -k 10 300 python -u tools/esa_bench.py --check 64 --ctu > gpurun_out/${TAG}_esa_bench.txt 2>&1
timeout -k 10 300 python -u tools/esa_bench.py --depth 10 --check 32 --ctu > gpurun_out/${TAG}_esa_main10_bench.txt 2>&1
timeout -k 10 300 python -u tools/rk_bench.py > gpurun_out/${TAG}_rk_bench.txt 2>&1
timeout -k 10 300 python -u tools/la_bench.py > gpurun_out/${TAG}_la_bench.txt 2>&1
for m in star hex dia umh; do timeout -k 10 200 python -u tools/pattern_bench.py --method $m >> gpurun_out/${TAG}_pattern_4k.txt 2>&1; done
timeout -k 10 300 python -u tools/rect_bench.py --pairs 16 --check 6 > gpurun_out/${TAG}_rect_bench.txt 2>&1
for d in 8 10; do timeout -k 10 100 python -u tools/mixed_bench.py --depth $d >> gpurun_out/${TAG}_mixed_bench.txt 2>&1; done
for p in 32 8 1; do GPU_MAX_HW_QUEUES=8 timeout -k 10 100 python -u tools/rows_bench.py --pairs $p >> gpurun_out/${TAG}_rows_bench.txt 2>&1; done
bash tools/profile_round.sh $TAG
