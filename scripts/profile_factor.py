"""developer script: refactor kernel time with parts disabled (results invalid, timing only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import osqp_solver_amd as M
from osqp_solver_amd import problems as PR
pr = PR.random_box_qp(int(os.environ.get("B", "1024")))
s = M.BatchSolver(pr["P"], pr["Px"], pr["q"], pr["A"], pr["Ax"], pr["l"], pr["u"])
for skip, name in ((0, "full"), (1, "no rank-1"), (2, "no general"), (3, "no updates"), (3 + 4 + 8, "no U/D/T"), (31, "assembly+zero only"), (16, "no scatter")):
    os.environ["MI_OSQP_FACTOR_SKIP"] = str(skip)      # honoured only by a diagnostic build: MI_OSQP_CXXFLAGS=-DMI_OSQP_DEBUG_BUILD python osqp-solver_amd/build.py --force
    def run():
        try: s.refactor_device()
        except M.MiOsqpError: pass          # (with parts disabled the inertia check of the dense tail fails: timing only)
    run(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3): run()
    torch.cuda.synchronize()
    print(f"{name:22s} {(time.perf_counter()-t)/3*1e3:7.2f} ms", flush=True)
