#!/bin/bash
# developer script (GPU box): waves per workgroup on the GOMP configs (barrier-bound small QPs: fewer waves = cheaper phases?)
for t in 512 256 128 64; do
  echo "== MI_OSQP_THREADS=$t"
  MI_OSQP_THREADS=$t timeout -k 10 120 python scripts/profile_gomp.py 2>&1 | grep -E "config|Error|error" | cut -c1-200
done
