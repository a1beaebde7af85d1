"""profiling target (scripts/profile_config5_pmc.sh): BASELINE config 5 at its literal size, 200 ADMM iterations."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import osqp_solver_amd as M
from osqp_solver_amd import problems as PR
pr = PR.grid_qp(int(os.environ.get("G", "316")))
s = M.BatchSolver(pr["P"], pr["Px"], pr["q"], pr["A"], pr["Ax"], pr["l"], pr["u"], max_iter=200)
st = s.stats()
torch.cuda.synchronize()
t = time.perf_counter(); info = s.solve(); t1 = time.perf_counter() - t
print({k: st[k] for k in ("n", "m", "N", "nnz_L", "nnz_KKT", "fwd_levels", "bwd_levels", "fwd_slots", "bwd_slots", "chk_slots", "solve_groups", "solve_group_threads")})
print(f"{info[0].iter} iterations in {1e3 * t1:.1f} ms = {1e3 * t1 / info[0].iter:.3f} ms per iteration, rho updates {info[0].rho_updates}")
