"""Cost of one SQP re-linearisation step on the GOMP batch of config 4: update (new A values + bounds) against solve.
   MI_OSQP_DEBUG_TIMING=1 python scripts/update_probe.py"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
M = importlib.import_module("osqp-solver_amd")
PR = importlib.import_module("osqp-solver_amd.problems")
for Bq in (256, 1024):
    pr = PR.gomp_batch(Bq, 7, 100)
    s = M.BatchSolver(pr["P"], pr["Px"], None, pr["A"], pr["Ax"], pr["l"], pr["u"])
    s.warm_start_x(pr["warm"]); s.solve()
    rng = np.random.default_rng(1)
    tu, tsv = [], []
    for k in range(4):
        Ax2 = pr["Ax"] * (1.0 + 0.01 * rng.standard_normal(pr["Ax"].shape))
        torch.cuda.synchronize(); t = time.perf_counter()
        s.update_A_bounds(Ax2, pr["l"], pr["u"])
        torch.cuda.synchronize(); tu.append(time.perf_counter() - t)
        t = time.perf_counter(); info = s.solve(); tsv.append(time.perf_counter() - t)
    td = []
    for k in range(4):
        tA = torch.from_numpy(pr["Ax"] * (1.0 + 0.01 * rng.standard_normal(pr["Ax"].shape))).cuda()
        tl, tu_ = torch.from_numpy(pr["l"]).cuda(), torch.from_numpy(pr["u"]).cuda()
        torch.cuda.synchronize(); t = time.perf_counter()
        s.update_A_bounds_device(tA, tl, tu_)
        torch.cuda.synchronize(); td.append(time.perf_counter() - t)
        s.solve()
    print(f"B={Bq}: update {1e3 * min(tu):.2f} ms from host pointers, {1e3 * min(td):.2f} ms from device-resident values; solve {1e3 * min(tsv):.2f} ms ({max(i.iter for i in info)} iterations)", flush=True)
