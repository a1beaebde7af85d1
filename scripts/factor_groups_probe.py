"""developer probe: workgroups sharing the refactorisation of ONE QP (MI_OSQP_FACTOR_GROUPS).   python scripts/factor_groups_probe.py"""
import importlib, os, subprocess, sys, time
if os.environ.get("FG_CHILD"):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    M = importlib.import_module("osqp-solver_amd")
    PR = importlib.import_module("osqp-solver_amd.problems")
    which = os.environ["FG_CHILD"]
    if which == "grid316": pr = PR.grid_qp(316); q = pr["q"]
    elif which == "grid150": pr = PR.grid_qp(150); q = pr["q"]
    else: pr = PR.gomp_batch(1, 6, int(which)); q = None
    s = M.BatchSolver(pr["P"], pr["Px"], q, pr["A"], pr["Ax"], pr["l"], pr["u"])
    ts = []
    for _ in range(4):
        torch.cuda.synchronize(); t = time.perf_counter(); s.refactor_device(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    print(f"{which}: factor groups {os.environ.get('MI_OSQP_FACTOR_GROUPS', 'default')}: refactor_device {1e3 * min(ts):.3f} ms (N = {s.stats()['N']})", flush=True)
    sys.exit(0)
for which in ("50", "802", "grid150", "grid316"):
    for g in ("default", "2", "4", "8", "16", "32", "64", "128"):
        env = dict(os.environ, FG_CHILD=which)
        if g != "default": env["MI_OSQP_FACTOR_GROUPS"] = g
        subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, timeout=300)
