#!/bin/bash
# One real encode of BASELINE configs[0] (416x240, 30 frames, --preset medium --me hex --threads 1) by the reference
# encoder with its motion estimation answered by the engine (oracle/_ref/x265_served), in check mode and in gpu mode,
# then the same binary on the CPU alone. Prints bitstream MD5s, fps lines and the binding's statistics.
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/enc
python3 - <<'PY'
from x265_amd.synth import baseline_clip, write_yuv420
write_yuv420("gpurun_out/enc/c0.yuv", baseline_clip(416, 240, 30, 1234))
PY
B=oracle/_ref/x265_served
ARGS="--input gpurun_out/enc/c0.yuv --input-res 416x240 --fps 30 -f 30 --preset medium --me hex --threads 1 --no-progress"
for m in ${MODES:-check gpu cpu}; do
  if [ $m = cpu ]; then unset X265AMD_SERVE; else export X265AMD_SERVE=$m; fi
  X265AMD_SERVE_STATS=gpurun_out/enc/stats_$m.json timeout -k 10 300 $B $ARGS -o gpurun_out/enc/c0_$m.hevc > gpurun_out/enc/log_$m.txt 2>&1
  echo "== $m: $(md5sum < gpurun_out/enc/c0_$m.hevc) $(grep encoded gpurun_out/enc/log_$m.txt)"
  cat gpurun_out/enc/stats_$m.json
done
rm -f gpurun_out/enc/c0.yuv gpurun_out/enc/*.hevc
