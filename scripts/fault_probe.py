"""Diagnostic build only (tests/test_gpu_faults.py runs this with MI_OSQP_LIBRARY = libmi_osqp_debug.so): a workgroup of a
grid-spinning launch never shows up.  Every such launch - the dataflow iterate / check / KKT-solve kernels of a large single
QP, the refactorisation shared by a group of workgroups - must end with MI_OSQP_ERR_DEVICE after the 2 s time-out, and the
handle must be usable afterwards WITHOUT a reset(): a failed solve puts it back into its last good state (cold start), a
failed refactorisation leaves the QP without a valid factor (kNonConvex) until the next refactorisation succeeds.
Prints one line per check; the last line is FAULTS OK or FAULTS FAILED."""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

M = importlib.import_module("osqp-solver_amd")
PR = importlib.import_module("osqp-solver_amd.problems")
fails = 0


def check(ok, what):
    global fails
    print(("ok   " if ok else "FAIL ") + what, flush=True)
    fails += 0 if ok else 1


def expect_device_error(fn, what):
    t = time.time()
    try:
        fn()
        check(False, what + ": no error raised")
    except M.MiOsqpError as e:
        dt = time.time() - t
        check(e.code == 5 and dt < 6.0, "%s: error %d after %.1f s" % (what, e.code, dt))


pr = PR.grid_qp(90)
s = M.BatchSolver(pr["P"], pr["Px"], pr["q"], pr["A"], pr["Ax"], pr["l"], pr["u"])
assert s.stats()["solve_groups"] > 1
ok = s.solve()[0]
x_ok = s.primal().copy()
check(ok.exit_code == 0, "healthy solve: %d iterations" % ok.iter)
for which in ("iterate", "check"):
    s.reset()
    os.environ["MI_OSQP_DEBUG_DROP_GROUP"] = which
    expect_device_error(s.solve, "solve with a missing workgroup in " + which)
    os.environ.pop("MI_OSQP_DEBUG_DROP_GROUP")
    again = s.solve()[0]                         # no reset(): the failed solve left the last good state, cold-started
    check((again.iter, again.exit_code) == (ok.iter, ok.exit_code) and np.array_equal(s.primal(), x_ok),
          "next solve after the %s fault, no reset: %d iterations, bitwise the healthy one" % (which, again.iter))
rhs = torch.tensor(np.random.default_rng(3).standard_normal((1, s.stats()["N"])), device="cuda")
sol = torch.empty_like(rhs)
s.kkt_solve_device(rhs, sol)
sol_ok = sol.clone()
os.environ["MI_OSQP_DEBUG_DROP_GROUP"] = "kkt"
expect_device_error(lambda: s.kkt_solve_device(rhs, sol), "KKT-solve op with a missing workgroup")
os.environ.pop("MI_OSQP_DEBUG_DROP_GROUP")
s.kkt_solve_device(rhs, sol)
check(bool(torch.equal(sol, sol_ok)), "KKT-solve op after the fault: bitwise the healthy one")
# the refactorisation shared by a group of workgroups (a lone QP's rho update / refactor_device)
os.environ["MI_OSQP_DEBUG_DROP_GROUP"] = "factor"
expect_device_error(s.refactor_device, "refactorisation with a missing workgroup")
os.environ.pop("MI_OSQP_DEBUG_DROP_GROUP")
bad = s.solve()[0]
check(bad.exit_code == 9 and bool(np.all(np.isnan(s.primal()))), "solve after the failed refactorisation: %s (no valid factor)" % M.EXIT_NAMES[bad.exit_code])
s.refactor_device()
s.reset()
again = s.solve()[0]
check((again.iter, again.exit_code) == (ok.iter, ok.exit_code), "after a successful refactorisation: %d iterations, %s" % (again.iter, M.EXIT_NAMES[again.exit_code]))
print("FAULTS FAILED (%d)" % fails if fails else "FAULTS OK", flush=True)
sys.exit(1 if fails else 0)
