# Main10 paths: parity tests of both depths, then the 4K Main10 / 8-bit pattern search timings
set -e
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests/test_gpu_search.py tests/test_gpu_configs.py -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3
cat > gpurun_out/m.py <<'PY'
import torch
from x265_amd.engine import Engine
from x265_amd import measure
dev = torch.device("cuda", 0)
e10 = Engine(0, 10)
m = measure.pattern_search_picture(e10, dev)
print("main10 4K star subme5: %.3f ms per picture-reference, %.1f Gpixel-SAD/s" % (m["ms"], m["pixel_sads"] / m["ms"] / 1e6))
e8 = Engine(0, 8)
m = measure.pattern_search_picture(e8, dev, depth=8, qp=32)
print("8-bit 4K star subme5: %.3f ms per picture-reference, %.1f Gpixel-SAD/s" % (m["ms"], m["pixel_sads"] / m["ms"] / 1e6))
PY
timeout -k 10 300 python gpurun_out/m.py
