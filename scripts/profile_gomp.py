"""developer script: GOMP configs 2 and 4 -- solve time GPU vs oracle on this box's cores."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import osqp_solver_amd as M
from osqp_solver_amd import problems as PR
from oracle import oracle as O
for name, B, D, W in (("config2", 1, 6, 50), ("config4", 256, 7, 100)):
    pr = PR.gomp_batch(B, D, W)
    t = time.time(); s = M.BatchSolver(pr["P"], pr["Px"], None, pr["A"], pr["Ax"], pr["l"], pr["u"]); ts = time.time() - t
    st = s.stats()
    s.warm_start_x(pr["warm"]); s.solve()
    ts_ = []
    for k in range(3):
        s.reset(); s.warm_start_x(pr["warm"]); torch.cuda.synchronize()
        t = time.perf_counter(); info = s.solve(); ts_.append(time.perf_counter() - t)
    its = np.array([i.iter for i in info]); ls = s.last_solve_stats()
    cores = M.host_cores()
    nb = min(B, 4 * cores)
    r = O.batch_solve(pr["P"], pr["Px"][:nb], None, pr["A"], pr["Ax"][:nb], pr["l"][:nb], pr["u"][:nb], threads=cores, native=True)
    print(f"{name}: B={B} N={st['N']} nnzL={st['nnz_L']} phases={st['fwd_levels']}+{st['bwd_levels']} tile={st['tile']} setup {ts:.2f}s | "
          f"GPU solve {min(ts_)*1e3:.2f} ms ({B/min(ts_):.0f} QPs/s) iters mean {its.mean():.0f} max {its.max()} {ls} | "
          f"oracle cold-start {nb} QPs on {cores} threads: {nb/r['solve_s']:.0f} QPs/s iters mean {r['iters'].mean():.0f}", flush=True)
