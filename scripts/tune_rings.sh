#!/bin/bash
# developer script (GPU box): ring depths (compile-time) x tile shapes; rebuilds the library per variant
run() { python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   QPs/s %.0f  ms/step %.2f' % (d['value'], d['ms_per_step']), {k: round(v,1) for k,v in d['step_breakdown_ms'].items() if k in ('device_iterate','device_refactor')}, 'tile', d['config']['tile'])"; }
for flags in "" "-DMI_PFV=6 -DMI_DT_PF=8" "-DMI_PFV=9 -DMI_DT_PF=8" "-DMI_PFV=9 -DMI_DT_PF=16"; do
  echo "== flags: $flags"
  MI_OSQP_CXXFLAGS="$flags" python osqp-solver_amd/build.py --force > /dev/null 2>&1 || { echo build failed; continue; }
  for t in 2 1; do echo "  tile $t"; MI_OSQP_TILE=$t run; done
done
MI_OSQP_CXXFLAGS="" python osqp-solver_amd/build.py --force > /dev/null 2>&1
