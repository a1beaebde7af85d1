import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import osqp_solver_amd as M
from osqp_solver_amd import problems as PR
B = int(os.environ.get("B", "1024"))
pr = PR.random_box_qp(B)
s = M.BatchSolver(pr["P"], pr["Px"], pr["q"], pr["A"], pr["Ax"], pr["l"], pr["u"])
n, m = pr["n"], pr["m"]
rhs = torch.randn(B, n + m, dtype=torch.float64, device="cuda"); sol = torch.empty_like(rhs)
for _ in range(3): s.kkt_solve_device(rhs, sol)
torch.cuda.synchronize(); t0 = time.perf_counter()
R = 50
for _ in range(R): s.kkt_solve_device(rhs, sol)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / R
st = s.stats()
byt = (st["fwd_slots"] + st["bwd_slots"] + st["dense_tail_slots"]) * 8 * B
print(f"B={B} kkt_solve {dt*1e6:.1f} us/call (host-timed, incl. launch+sync), streamed {byt/1e9:.3f} GB -> {byt/dt/1e12:.2f} TB/s")
s.close()
