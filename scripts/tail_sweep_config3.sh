#!/bin/bash
# GPU box: forced dense-tail sizes on the headline batch (ms per step, device iterate / refactor split)
cd "$(dirname "$0")/.."
for k in 256 320 384 448 512; do
  echo "== MI_OSQP_DENSE_TAIL=$k"
  MI_OSQP_DENSE_TAIL=$k timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['step_breakdown_ms'], d['config'].get('dense_tail_rows'), d['config'].get('nnz_L_before_tail'), d['config'].get('fwd_levels'))"
done
