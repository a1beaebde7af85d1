"""Multi-workgroup mode probe (one large QP shared by G workgroups): results against G = 1, time per iteration.
   python scripts/mw_probe.py [G[:threads] ...]"""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
M = importlib.import_module("osqp-solver_amd")
PR = importlib.import_module("osqp-solver_amd.problems")
Gs = sys.argv[1:] or ["0", "16", "32", "64:256", "128:128"]        # workgroups[:threads per workgroup]; 0 = barrier form
for g in (150, 316):
    pr = PR.grid_qp(g)
    base = None
    for G in Gs:
        os.environ["MI_OSQP_GROUPS"] = G.split(":")[0]
        os.environ["MI_OSQP_GROUP_THREADS"] = G.split(":")[1] if ":" in G else "512"
        s = M.BatchSolver(pr["P"], pr["Px"], pr["q"], pr["A"], pr["Ax"], pr["l"], pr["u"], max_iter=200)
        st = s.stats()
        rf = s.refactor_time()
        torch.cuda.synchronize()
        t = time.perf_counter(); info = s.solve(); t1 = time.perf_counter() - t
        x = s.primal()[0].copy()
        rhs = np.random.default_rng(1).standard_normal((1, st["N"]))
        d_rhs = torch.tensor(rhs, device="cuda"); d_sol = torch.empty_like(d_rhs)
        s.kkt_solve_device(d_rhs, d_sol)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5): s.kkt_solve_device(d_rhs, d_sol)
        torch.cuda.synchronize(); tk = (time.perf_counter() - t) / 5
        sol = d_sol.cpu().numpy()[0]
        if base is None: base = (x, sol, info[0].iter)
        print(f"g={g} G={G} iters {info[0].iter} status {info[0].exit_code} ms/iter {1e3 * t1 / max(1, info[0].iter):.3f} kkt_solve {1e3 * tk:.3f} ms "
              f"dx {np.max(np.abs(x - base[0])):.2e} dsol {np.max(np.abs(sol - base[1])) / np.max(np.abs(base[1])):.2e} levels {st['fwd_levels']}+{st['bwd_levels']} setup factor_kernel {rf[0]:.1f} ms (MI_OSQP_FACTOR_GROUPS={os.environ.get('MI_OSQP_FACTOR_GROUPS', 'default')})", flush=True)
        s.close()
