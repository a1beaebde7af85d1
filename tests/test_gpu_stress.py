"""Randomized GPU-vs-oracle sweeps (fixed seeds) as part of the GPU suite: the developer scripts
scripts/stress_random.py (shapes, tilings, thread counts, LDS / global solve vector) and
scripts/stress_settings.py (solver settings and call sequences) exit non-zero on any mismatch in exit code,
iteration count, number of rho updates or x; scripts/stress_dense_tail.py does the same for the dense tail under bad
conditioning (nearly-LP objectives, equality and free rows, tight tolerances; chosen and forced tails)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("script,env", [("stress_random.py", {"TRIALS": "24"}),
                                        ("stress_settings.py", {"TRIALS": "40", "SEED": "11"}),
                                        ("stress_settings.py", {"TRIALS": "40", "SEED": "23"}),
                                        ("stress_dense_tail.py", {"TRIALS": "12"}),
                                        ("stress_dense_tail.py", {"TRIALS": "10", "SEED": "9", "FORCE": "256"})])
def test_randomized_parity_sweep(script, env):
    e = dict(os.environ); e.update(env)
    for k in ("MI_OSQP_TILE", "MI_OSQP_THREADS", "MI_OSQP_GLOBAL_XS", "MI_OSQP_DENSE_TAIL"): e.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", script)], capture_output=True, text=True, timeout=900, env=e)
    assert r.returncode == 0 and "0 problems" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
