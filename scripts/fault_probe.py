"""developer check (diagnostic build only): a workgroup of the dataflow grid never shows up - the waits must time out, the
solve must return MI_OSQP_ERR_DEVICE within seconds, and the GPU must be usable afterwards.
   MI_OSQP_CXXFLAGS=-DMI_OSQP_DEBUG_BUILD python osqp-solver_amd/build.py --force; python scripts/fault_probe.py"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
M = importlib.import_module("osqp-solver_amd")
PR = importlib.import_module("osqp-solver_amd.problems")
pr = PR.grid_qp(90)
s = M.BatchSolver(pr["P"], pr["Px"], pr["q"], pr["A"], pr["Ax"], pr["l"], pr["u"])
ok = s.solve()[0]
print("healthy solve:", ok.iter, ok.exit_code, flush=True)
os.environ["MI_OSQP_DEBUG_DROP_GROUP"] = "1"
s.reset()
t = time.time()
try:
    s.solve(); print("NO ERROR RAISED", flush=True)
except M.MiOsqpError as e:
    print("error after %.1f s: %s" % (time.time() - t, e), flush=True)
os.environ.pop("MI_OSQP_DEBUG_DROP_GROUP")
s.reset()
again = s.solve()[0]
print("solve after the fault:", again.iter, again.exit_code, "same as before:", (again.iter, again.exit_code) == (ok.iter, ok.exit_code), flush=True)
