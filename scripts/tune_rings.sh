#!/bin/bash
# developer script (GPU box): ring depths (compile-time) of the sweeps / the dense-tail product on the headline batch
# (one QP per tile, 16 waves: MI_PFV_LAT is its ring); rebuilds the library per variant in the box's scratch copy.
#   bash scripts/tune_rings.sh
run() { python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   QPs/s %.0f  ms/step %.2f' % (d['value'], d['ms_per_step']), {k: round(v,1) for k,v in d['step_breakdown_ms'].items() if k in ('device_iterate','device_refactor')}, d['roofline']['kernel'])"; }
variants=("" "-DMI_PFV_LAT=9" "" "-DMI_PFV_LAT=12" "-DMI_DT_PF=16" "-DMI_PFV_LAT=3" "")      # (the default in between: boxes drift by 2-3 %)
for flags in "${variants[@]}"; do
  echo "== flags: $flags"
  MI_OSQP_CXXFLAGS="$flags" python osqp-solver_amd/build.py --force > /dev/null 2>&1 || { echo build failed; continue; }
  run; run
done
