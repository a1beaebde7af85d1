import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import osqp_solver_amd as M
from osqp_solver_amd import problems as PR
from oracle import oracle as O
B = 4
pr = PR.random_box_qp(B)
s = M.BatchSolver(pr["P"], pr["Px"], pr["q"], pr["A"], pr["Ax"], pr["l"], pr["u"])
print({k: v for k, v in s.stats().items() if "dense" in k or "slots" in k or "levels" in k})
n, m = pr["n"], pr["m"]
rng = np.random.default_rng(0)
rhs = rng.standard_normal((B, n + m))
d_rhs = torch.tensor(rhs, device="cuda"); d_sol = torch.empty_like(d_rhs)
s.kkt_solve_device(d_rhs, d_sol)
sol = d_sol.cpu().numpy()
for q in range(B):
    P, A = PR.qp_matrices(pr, q)
    o = O.OracleQPSolver(P, pr["q"][q], A, pr["l"][q], pr["u"][q])
    ref = o.kkt_solve(rhs[q])
    print(q, "kkt err", np.abs(sol[q] - ref).max() / np.abs(ref).max())
s.close()
