"""GPU (-m gpu): the time-out paths of the grid-spinning launches, by fault injection in the diagnostic twin of the library
(osqp-solver_amd/libmi_osqp_debug.so, built by build.build_debug with -DMI_OSQP_DEBUG_BUILD; the product binary has no such
switch).  scripts/fault_probe.py runs in a child process that loads the twin through MI_OSQP_LIBRARY."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_missing_workgroup_ends_in_an_error_and_the_handle_recovers():
    sys.path.insert(0, ROOT)
    import importlib.util
    spec = importlib.util.spec_from_file_location("_mi_build", os.path.join(ROOT, "osqp-solver_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    lib = mod.build_debug()
    env = dict(os.environ, MI_OSQP_LIBRARY=lib)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fault_probe.py")], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "FAULTS OK" in r.stdout, r.stdout[-4000:] + r.stderr[-2000:]
