#!/bin/bash
# developer script: bench under different (tile, threads-per-workgroup) shapes
for cfg in "2 512" "1 512" "4 512" "2 256"; do
  set -- $cfg
  echo "== tile=$1 threads=$2"
  MI_OSQP_TILE=$1 MI_OSQP_THREADS=$2 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('QPs/s %.0f  ms/step %.1f  iterate %.1f refactor %.1f  roofline %.0f GB/s' % (d['value'], d['ms_per_step'], d['step_breakdown_ms']['device_iterate'], d['step_breakdown_ms']['device_refactor'], d['roofline']['achieved']))"
done
