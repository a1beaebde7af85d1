#!/bin/bash
# developer script (GPU box): forced dense tails on the GOMP configs (latency-bound single QPs: fewer phases?)
for k in 0 64 128 192 256 320; do
  echo "== MI_OSQP_DENSE_TAIL=$k"
  MI_OSQP_DENSE_TAIL=$k timeout -k 10 120 python scripts/profile_gomp.py 2>&1 | grep -E "config|Error|error" | cut -c1-330
done
