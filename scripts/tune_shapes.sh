#!/bin/bash
# developer script: bench under different (tile, threads-per-workgroup) shapes
for cfg in "4 512 0" "2 512 0" "4 512 1" "2 512 1" "1 512 0"; do
  set -- $cfg
  echo "== tile=$1 threads=$2 no_compact=$3"
  if [ "$3" = "1" ]; then unset MI_OSQP_COMPACT; else export MI_OSQP_COMPACT=1; fi   # (compaction is opt-in: MI_OSQP_COMPACT=1; third argument 1 = off)
  MI_OSQP_TILE=$1 MI_OSQP_THREADS=$2 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('QPs/s %.0f  ms/step %.1f' % (d['value'], d['ms_per_step']), {k: round(v,1) for k,v in d['step_breakdown_ms'].items()}, 'roofline %.0f' % d['roofline']['achieved'])"
done
