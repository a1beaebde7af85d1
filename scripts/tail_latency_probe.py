"""developer probe: a forced dense tail (MI_OSQP_DENSE_TAIL=k) on the chain-like GOMP factors in the latency-bound regime
(one QP per CU or fewer): phases per iteration and ms per 25-iteration solve.   python scripts/tail_latency_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import osqp_solver_amd as M
from osqp_solver_amd import problems as PR

for name, B, D, W in (("config 2 (1 x 6 x 50)", 1, 6, 50), ("3-DOF x 100", 1, 3, 100), ("3-DOF x 60", 1, 3, 60), ("256 x 3 x 60", 256, 3, 60), ("256 x 7 x 100", 256, 7, 100)):
    pr = PR.gomp_batch(B, D, W)
    for k in os.environ.get("KS", "0,64,128,192,256").split(","):
        os.environ["MI_OSQP_DENSE_TAIL"] = k
        try:
            s = M.BatchSolver(pr["P"], pr["Px"], None, pr["A"], pr["Ax"], pr["l"], pr["u"])
        except M.MiOsqpError as e:
            print(name, "k", k, "setup failed:", e); continue
        st = s.stats()
        s.warm_start_x(pr["warm"]); info = s.solve()
        ts = []
        for _ in range(7):
            s.reset(); s.warm_start_x(pr["warm"]); torch.cuda.synchronize()
            t = time.perf_counter(); info = s.solve(); ts.append(time.perf_counter() - t)
        kt, nl = s.kernel_time()
        t0 = time.perf_counter(); s.refactor_device(); tr = time.perf_counter() - t0
        its = max(i.iter for i in info)
        print(f"{name:22s} k={st['dense_tail_rows']:3d} N={st['N']} phases {st['fwd_levels']}+{st['bwd_levels']} solve {1e3*min(ts):.3f} ms ({its} it), "
              f"iterate kernel {kt:.3f} ms per launch = {1e3*kt/25:.1f} us/it, refactor call {1e3*tr:.2f} ms", flush=True)
        s.close()
