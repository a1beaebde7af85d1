"""developer script: the dense tail under bad conditioning - nearly-LP objectives (P scaled down to 1e-6), equality rows
(rho x 1000), free rows (rho_min), tight tolerances - GPU vs oracle on exit code, iteration count and x."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import osqp_solver_amd as M
from osqp_solver_amd import problems as PR
from oracle import oracle as O
ST2EXIT = {1: 0, 2: 3, -3: 1, 3: 4, -4: 2, 4: 5, -2: 6, -7: 9, -10: 10}
rng = np.random.default_rng(int(os.environ.get("SEED", "3")))
bad = cases = with_tail = 0
for trial in range(int(os.environ.get("TRIALS", "16"))):
    n = int(rng.integers(150, 500)); mg = int(rng.integers(150, 500)); nnz = int(rng.integers(4, 10)); B = int(rng.integers(1, 4))
    pr = PR.random_box_qp(B, n=n, mg=mg, nnz_per_row=nnz, pattern_seed=int(rng.integers(1 << 30)))
    m = pr["m"]
    pscale = float(rng.choice([1.0, 1e-3, 1e-6]))
    pr["Px"] = pr["Px"] * pscale
    x0 = rng.uniform(-0.05, 0.05, (B, n))
    kind = rng.random(m)
    for b in range(B):
        P, A = PR.qp_matrices(pr, b)
        ax0 = A @ x0[b]
        eq = kind < 0.15; free = (kind >= 0.15) & (kind < 0.35)
        pr["l"][b][eq] = ax0[eq]; pr["u"][b][eq] = ax0[eq]
        pr["l"][b][free] = -1e30; pr["u"][b][free] = 1e30
    eps = float(rng.choice([1e-3, 1e-5]))
    kw = dict(eps_abs=eps, eps_rel=eps, max_iter=int(rng.choice([4000, 600])))
    if os.environ.get("FORCE") and n + m > 600: os.environ["MI_OSQP_DENSE_TAIL"] = os.environ["FORCE"]
    else: os.environ.pop("MI_OSQP_DENSE_TAIL", None)
    s = M.BatchSolver(pr["P"], pr["Px"], pr["q"], pr["A"], pr["Ax"], pr["l"], pr["u"], **kw)
    k = s.stats()["dense_tail_rows"]; with_tail += 1 if k else 0
    info = s.solve(); x = s.primal()
    for b in range(B):
        P, A = PR.qp_matrices(pr, b)
        o = O.OracleQPSolver(P, pr["q"][b], A, pr["l"][b], pr["u"][b], **kw)
        st, xo = o.solve(); io = o.info()
        cases += 1
        has = st in (1, 2, -2)
        dx = float(np.nanmax(np.abs(x[b] - xo))) if has else 0.0
        if info[b].exit_code != ST2EXIT[st] or info[b].iter != io.iter or dx > 1e-6 * max(1.0, float(np.max(np.abs(xo))) if has else 1.0):
            bad += 1
            print("trial", trial, "MISMATCH qp", b, dict(n=n, m=m, nnz=nnz, pscale=pscale, tail=k, **kw), "gpu", info[b].exit_code, info[b].iter, "oracle", ST2EXIT[st], io.iter, "dx", dx)
    s.close()
print(f"{cases} QPs compared, {bad} problems, {with_tail} batches with a dense tail")
sys.exit(1 if bad else 0)
