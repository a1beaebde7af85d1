"""developer script: BASELINE config 5 (single large structured sparse QP) through the wide-index path.
G=316 is the literal size (n = 99 856, m = 298 936)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import osqp_solver_amd as M
from osqp_solver_amd import problems as PR
from oracle.kkt_check import kkt_residuals
g = int(os.environ.get("G", "316"))
pr = PR.grid_qp(g)
t = time.time()
s = M.BatchSolver(pr["P"], pr["Px"], pr["q"], pr["A"], pr["Ax"], pr["l"], pr["u"])
ts = time.time() - t
st = s.stats()
print({k: st[k] for k in ("n", "m", "N", "nnz_L", "fwd_levels", "bwd_levels", "fwd_slots", "bwd_slots", "tile", "threads_per_block")}, f"setup {ts:.1f} s")
t = time.time(); info = s.solve(); t1 = time.time() - t
print(f"solve: {t1*1e3:.0f} ms, {info[0].iter} iterations, exit {info[0].exit_code}, {t1*1e3/max(1, info[0].iter):.2f} ms per iteration, rho updates {info[0].rho_updates}")
x, y = s.primal()[0], s.dual()[0]
P, A = PR.qp_matrices(pr, 0)
print("kkt", kkt_residuals(P, pr["q"][0], A, pr["l"][0], pr["u"][0], x, y))
if os.environ.get("ORACLE"):
    from oracle import oracle as O
    t = time.time(); o = O.OracleQPSolver(P, pr["q"][0], A, pr["l"][0], pr["u"][0]); to = time.time() - t
    t = time.time(); sto, xo = o.solve(); t2 = time.time() - t
    print(f"oracle: setup {to:.1f} s, solve {t2*1e3:.0f} ms, {o.info().iter} iterations, status {sto}; max |x - x_oracle| = {np.abs(x - xo).max():.2e}")
s.close()
