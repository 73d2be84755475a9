#!/bin/bash
# Experiment: how the search launches of bench.py split between launch modes / methods / sub-pel levels.
# Prints the timed launch class's duration and the step time; run on the GPU box.
set -e
for cfg in "$@"; do
  set -- $cfg
  timeout -k 10 120 python bench.py --steps 3 --warmup 1 --pairs 8 --cpu-pairs 0 --no-other-paths --method $1 --subme $2 --launch $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1 $2 $3', 'launch_ms=%.3f'%d['roofline']['launch_ms'], 'ms_per_step=%.2f'%d['ms_per_step'], 'value=%.0f'%d['value'])"
done
