"""developer script: randomized parity sweep (GPU vs oracle) over SETTINGS and call sequences:
rho / sigma / alpha, scaling passes, adaptive rho on/off and interval, check interval, max_iter, tolerances,
warm starts, A-value updates, repeated solves; random box QPs, GOMP batches and infeasible / unbounded variants."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import osqp_solver_amd as M
from osqp_solver_amd import problems as PR
from oracle import oracle as O

ST2EXIT = {1: 0, 2: 3, -3: 1, 3: 4, -4: 2, 4: 5, -2: 6, -7: 9, -10: 10}
rng = np.random.default_rng(int(os.environ.get("SEED", "11")))
bad = cases = 0
for trial in range(int(os.environ.get("TRIALS", "40"))):
    kind = rng.choice(["box", "box", "gomp", "infeasible", "unbounded"])
    B = int(rng.integers(1, 6))
    if kind == "gomp":
        pr = PR.gomp_batch(B, int(rng.integers(2, 5)), int(rng.integers(6, 20)), seed=int(rng.integers(1 << 20)))
    else:
        n = int(rng.integers(4, 120)); mg = int(rng.integers(1, 140)); nnz = int(rng.integers(1, min(n, 8) + 1))
        pr = PR.random_box_qp(B, n=n, mg=mg, nnz_per_row=nnz, pattern_seed=int(rng.integers(1 << 30)))
        if kind == "infeasible":                     # contradictory box on the first variable of half of the QPs
            pr["l"] = pr["l"].copy(); pr["u"] = pr["u"].copy()
            for b in range(0, B, 2):
                pr["l"][b, 0] = 2.0; pr["u"][b, 0] = 3.0
                rows = np.nonzero(pr["A"].tocsr()[:, 0].toarray().ravel())[0] if False else []
        if kind == "unbounded":                      # remove the boxes and the curvature of one direction: q'x -> -inf
            pr["l"] = np.full_like(pr["l"], -1e30); pr["u"] = np.full_like(pr["u"], 1e30)
    kw = {}
    if rng.random() < 0.5: kw["rho"] = float(10 ** rng.uniform(-2, 1))
    if rng.random() < 0.3: kw["sigma"] = float(10 ** rng.uniform(-7, -4))
    if rng.random() < 0.4: kw["alpha"] = float(rng.uniform(1.0, 1.9))
    if rng.random() < 0.4: kw["scaling"] = int(rng.choice([0, 3, 10, 15]))
    if rng.random() < 0.3: kw["adaptive_rho"] = 0
    if rng.random() < 0.4: kw["adaptive_rho_interval"] = int(rng.choice([25, 50, 75, 100]))
    if rng.random() < 0.4: kw["check_termination"] = int(rng.choice([1, 5, 10, 25, 40]))
    if rng.random() < 0.3: kw["max_iter"] = int(rng.choice([30, 100, 250, 1000]))
    if rng.random() < 0.5: e = float(10 ** rng.uniform(-7, -2)); kw["eps_abs"] = e; kw["eps_rel"] = e
    if rng.random() < 0.2: kw["scaled_termination"] = 1
    os.environ["MI_OSQP_TILE"] = str(int(rng.choice([1, 2, 4])))
    # round 2: equilibration of the update path on the device / on the host; a single QP through the dataflow path
    os.environ.pop("MI_OSQP_DEVICE_RUIZ", None); os.environ.pop("MI_OSQP_HOST_RUIZ", None); os.environ.pop("MI_OSQP_GLOBAL_XS", None)
    os.environ["MI_OSQP_DEVICE_RUIZ" if rng.random() < 0.6 else "MI_OSQP_HOST_RUIZ"] = "1"
    if B == 1 and rng.random() < 0.5: os.environ["MI_OSQP_GLOBAL_XS"] = "1"; os.environ["MI_OSQP_TILE"] = "1"
    seq = rng.choice(["solve", "solve2", "warm", "updA", "updAB", "bounds"])
    try:
        s = M.BatchSolver(pr["P"], pr["Px"], pr["q"], pr["A"], pr["Ax"], pr["l"], pr["u"], **kw)
        Ax2 = pr["Ax"] * (1.0 + 0.05 * np.sin(np.arange(pr["Ax"].shape[1])))[None, :]
        x0 = 0.1 * np.cos(np.arange(pr["n"]))[None, :].repeat(B, 0)
        if seq == "warm": s.warm_start_x(x0)
        info = s.solve()
        if seq == "solve2": info = s.solve()
        if seq == "updA": s.update_A(Ax2); info = s.solve()
        if seq == "updAB": s.update_A_bounds(Ax2, pr["l"] - 0.05, pr["u"] + 0.05); info = s.solve()
        if seq == "bounds": s.update_bounds(pr["l"] - 0.05, pr["u"] + 0.05); info = s.solve()
        x = s.primal()
    except Exception as ex:
        print("trial", trial, "EXCEPTION", repr(ex), kind, seq, kw); bad += 1; continue
    for b in range(B):
        P, A = PR.qp_matrices(pr, b)
        qv = None if pr["q"] is None else pr["q"][b]
        try:
            o = O.OracleQPSolver(P, qv, A, pr["l"][b], pr["u"][b], **kw)
        except Exception as ex:
            print("trial", trial, "oracle setup EXCEPTION", repr(ex)); bad += 1; break
        if seq == "warm": o.set_warm_start(x0[b])
        st, xo = o.solve()
        if seq == "solve2": st, xo = o.solve()
        if seq == "updA":
            A2 = A.copy(); A2.data = Ax2[b].copy(); o.update(pr["l"][b], A2, pr["u"][b]); st, xo = o.solve()
        if seq == "updAB":
            A2 = A.copy(); A2.data = Ax2[b].copy(); o.update(pr["l"][b] - 0.05, A2, pr["u"][b] + 0.05); st, xo = o.solve()
        if seq == "bounds": o.update_bounds_only(pr["l"][b] - 0.05, pr["u"][b] + 0.05); st, xo = o.solve()
        io = o.info()
        tol = 1e-3 if st in (-2, 2) else 1e-6
        same_x = np.all(np.isnan(xo)) and np.all(np.isnan(x[b])) if np.any(np.isnan(xo)) else np.max(np.abs(x[b] - xo)) <= tol * (1 + np.max(np.abs(xo)))
        okk = info[b].exit_code == ST2EXIT[st] and info[b].iter == io.iter and info[b].rho_updates == io.rho_updates and same_x
        cases += 1
        if not okk:
            bad += 1
            print("trial", trial, "MISMATCH qp", b, kind, seq, kw, "tile", os.environ["MI_OSQP_TILE"], "shape", pr["n"], pr["m"],
                  "| gpu", info[b].exit_code, info[b].iter, info[b].rho_updates, "| oracle", ST2EXIT[st], io.iter, io.rho_updates)
    s.close()
print(f"{cases} QPs compared, {bad} problems")
sys.exit(1 if bad else 0)
