"""Import shim: the package directory is `osqp-solver_amd/` (not a valid Python
identifier), so `import osqp_solver_amd` resolves here and loads it."""
import importlib.util
import os
import sys

_d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "osqp-solver_amd")
_spec = importlib.util.spec_from_file_location(
    "osqp_solver_amd", os.path.join(_d, "__init__.py"), submodule_search_locations=[_d])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["osqp_solver_amd"] = _mod
_spec.loader.exec_module(_mod)
