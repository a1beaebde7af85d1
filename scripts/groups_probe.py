"""developer probe: the dataflow grid (MI_OSQP_GROUPS x MI_OSQP_GROUP_THREADS) on a mid-size single QP - the reference example's
802-waypoint trajectory QP (N = 43 284).   python scripts/groups_probe.py [G:threads ...]"""
import importlib, os, subprocess, sys, time
if os.environ.get("GROUPS_PROBE_CHILD"):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    M = importlib.import_module("osqp-solver_amd")
    PR = importlib.import_module("osqp-solver_amd.problems")
    pr = PR.gomp_batch(1, 6, int(os.environ.get("W", "802")))
    s = M.BatchSolver(pr["P"], pr["Px"], None, pr["A"], pr["Ax"], pr["l"], pr["u"])
    s.warm_start_x(pr["warm"]); s.solve()
    ts = []
    for _ in range(5):
        s.reset(); s.warm_start_x(pr["warm"]); torch.cuda.synchronize()
        t = time.perf_counter(); info = s.solve(); ts.append(time.perf_counter() - t)
    t0 = time.perf_counter(); s.refactor_device(); tr = time.perf_counter() - t0
    st = s.stats()
    print(f"groups {st['solve_groups']} x {st['solve_group_threads']} threads: solve {1e3 * min(ts):.3f} ms ({info[0].iter} it), refactor {1e3 * tr:.2f} ms, phases {st['fwd_levels']}+{st['bwd_levels']}", flush=True)
    sys.exit(0)
for G in (sys.argv[1:] or ["128:128", "64:128", "256:128", "128:64", "64:256", "32:256"]):
    g, t = G.split(":")
    subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, MI_OSQP_GROUPS=g, MI_OSQP_GROUP_THREADS=t, GROUPS_PROBE_CHILD="1"), timeout=300)
