"""CPU: pins the oracle (oracle/ is test infrastructure; parity officially
unpinned -- no reference outputs exist) against
  * closed-form fixtures and KKT-verified fixtures (tests/golden/qp_fixtures.json),
  * the iteration log upstream publishes for its demo QP,
  * solver-independent KKT conditions on random problems."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import oracle as O
from oracle.kkt_check import kkt_residuals
from osqp_solver_amd import problems as PR

NAMES = {v: k for k, v in O.STATUS.items()}


def test_upstream_documented_demo_log():
    """Known-answer test recalled from upstream's documentation page 'Setup and
    solve' (Python example, alpha=1.0; OSQP v0.6.0 banner):
        iter 1:  obj -4.9384e-03  pri 1.00e+00  dua 2.00e+02  rho 1.00e-01
        iter 50: obj  1.8800e+00  pri 1.91e-07  dua 7.50e-07  rho 1.38e+00
        status solved, 50 iterations, optimal rho estimate 1.36e+00
    That run adapted rho after 25 iterations (wall-clock rule), i.e. interval 25."""
    P = [[4, 1], [1, 2]]; q = [1, 1]; A = [[1, 1], [1, 0], [0, 1]]; l = [1, 0, 0]; u = [1, 0.7, 0.7]
    s = O.OracleQPSolver(P, q, A, l, u, alpha=1.0, max_iter=1)
    s.solve(); i = s.info()
    assert f"{i.obj_val:.4e}" == "-4.9384e-03"
    assert f"{i.pri_res:.2e}" == "1.00e+00" and f"{i.dua_res:.2e}" == "2.00e+02"
    s = O.OracleQPSolver(P, q, A, l, u, alpha=1.0, adaptive_rho_interval=25)
    st, x = s.solve(); i = s.info()
    assert st == 1 and i.iter == 50 and i.rho_updates == 1
    assert f"{i.obj_val:.4e}" == "1.8800e+00"
    assert f"{i.pri_res:.2e}" == "1.91e-07" and f"{i.dua_res:.2e}" == "7.50e-07"
    assert f"{i.rho:.2e}" == "1.38e+00" and f"{i.rho_estimate:.2e}" == "1.36e+00"
    np.testing.assert_allclose(x, [0.3, 0.7], atol=1e-6)


def test_fixtures_tight(qp_fixtures):
    for name, d in qp_fixtures.items():
        s = O.OracleQPSolver(d["P"], d["q"], d["A"], d["l"], d["u"], eps_abs=1e-9, eps_rel=1e-9, max_iter=200000)
        st, x = s.solve()
        assert O.STATUS[st] == d["status"], name
        if d["x"] is not None:
            np.testing.assert_allclose(x, d["x"], atol=2e-6, err_msg=name)
            np.testing.assert_allclose(s.y, d["y"], atol=2e-5, err_msg=name)
        else:
            assert np.all(np.isnan(x)), name      # store_solution: NaN when no solution


def test_default_eps_satisfies_kkt_to_tolerance():
    pr = PR.random_box_qp(3, n=40, mg=30, nnz_per_row=4)
    for b in range(3):
        P, A = PR.qp_matrices(pr, b)
        s = O.OracleQPSolver(P, pr["q"][b], A, pr["l"][b], pr["u"][b])
        st, x = s.solve(); i = s.info()
        assert st == 1 and i.iter % 25 == 0
        r = kkt_residuals(P, pr["q"][b], A, pr["l"][b], pr["u"][b], x, s.y)
        # termination: pri_res <= eps(1e-3)*(1 + norms), both residuals small
        assert r["prim"] < 5e-3 and r["stat"] < 5e-2
        s2 = O.OracleQPSolver(P, pr["q"][b], A, pr["l"][b], pr["u"][b], eps_abs=1e-9, eps_rel=1e-9, max_iter=100000)
        st2, x2 = s2.solve()
        assert st2 == 1 and np.max(np.abs(x - x2)) < 5e-2
        r2 = kkt_residuals(P, pr["q"][b], A, pr["l"][b], pr["u"][b], x2, s2.y)
        assert max(r2.values() if False else [r2["prim"], r2["stat"], r2["comp"]]) < 1e-6


def test_factor_reconstructs_kkt():
    """L D L' = perm(K): validates ordering + etree + numeric LDL' (row E5)."""
    P, (l, A, u), _ = PR.gomp_qp(3, 8, np.zeros(3), np.ones(3))
    s = O.OracleQPSolver(P, None, A, l, u, scaling=0)
    perm, Lp, Li, Lx, Dinv = s.factor()
    N = len(perm); n = A.shape[1]
    L = sp.csc_matrix((Lx, Li, Lp), shape=(N, N)) + sp.eye(N)
    K = (L @ sp.diags(1.0 / Dinv) @ L.T).toarray()
    Pu = sp.triu(sp.csc_matrix(P)).toarray(); Pf = Pu + np.triu(Pu, 1).T
    rho = np.where((l < -1e26) & (u > 1e26), 1e-6, np.where(u - l < 1e-4, 100.0, 0.1))
    Kn = np.block([[Pf + 1e-6 * np.eye(n), A.toarray().T], [A.toarray(), -np.diag(1.0 / rho)]])
    np.testing.assert_allclose(K, Kn[np.ix_(perm, perm)], atol=1e-9 * np.abs(Kn).max())
    assert (Dinv > 0).sum() == n                      # inertia (n, m)
    rhs = np.random.default_rng(0).standard_normal(N)
    np.testing.assert_allclose(Kn @ s.kkt_solve(rhs), rhs, atol=1e-6)


def test_setup_rejects_bad_input():
    P = sp.eye(2).tocsc(); A = sp.eye(2).tocsc()
    with pytest.raises(ValueError):
        O.OracleQPSolver(P, None, A, [1, 0], [0, 1])              # l > u
    with pytest.raises(ValueError):
        O.OracleQPSolver(-P, None, A, [0, 0], [1, 1])             # non-convex: wrong inertia
    with pytest.raises(ValueError):
        O.OracleQPSolver(P, None, A, [0, 0], [1, 1], alpha=2.5)   # invalid settings


def test_update_and_warm_start_sequence(qp_fixtures):
    """The reference's call pattern: ctor, setWarmStart, solve, update, solve
    ([REF] src/gomp-solver.h:61-88)."""
    d = qp_fixtures["generic_15x25"]
    s = O.OracleQPSolver(d["P"], d["q"], d["A"], d["l"], d["u"], eps_abs=1e-8, eps_rel=1e-8)
    s.set_warm_start(d["x"])
    st, x = s.solve()
    assert st == 1 and s.info().iter <= 400
    A2 = d["A"].copy(); A2[A2 != 0] *= 1.1
    s.update(d["l"] * 0.9, A2, d["u"] * 0.9)
    st, x2 = s.solve()
    assert st == 1
    fresh = O.OracleQPSolver(d["P"], d["q"], A2, d["l"] * 0.9, d["u"] * 0.9, eps_abs=1e-8, eps_rel=1e-8)
    _, xf = fresh.solve()
    np.testing.assert_allclose(x2, xf, atol=1e-5)
    A3 = d["A"].copy(); A3[0, :] = 1.0                            # pattern change must be refused
    with pytest.raises(ValueError):
        s.update(d["l"], A3, d["u"])
    with pytest.raises(ValueError):
        s.update(d["u"] + 1, d["A"], d["l"])                      # l > u


def test_max_iter_and_inaccurate_codes(qp_fixtures):
    d = qp_fixtures["generic_30x40"]
    s = O.OracleQPSolver(d["P"], d["q"], d["A"], d["l"], d["u"], eps_abs=1e-12, eps_rel=1e-12, max_iter=30)
    st, x = s.solve()
    assert O.STATUS[st] in ("kMaxIterations", "kOptimalInaccurate") and not np.any(np.isnan(x))


def test_batch_driver_matches_single():
    pr = PR.random_box_qp(4, n=30, mg=20, nnz_per_row=3)
    r = O.batch_solve(pr["P"], pr["Px"], pr["q"], pr["A"], pr["Ax"], pr["l"], pr["u"], threads=2)
    assert r["failed"] == 0
    for b in range(4):
        P, A = PR.qp_matrices(pr, b)
        s = O.OracleQPSolver(P, pr["q"][b], A, pr["l"][b], pr["u"][b])
        st, x = s.solve()
        assert st == r["status"][b] and s.info().iter == r["iters"][b]
        np.testing.assert_array_equal(x, r["x"][b])


def test_oracle_with_a_caller_supplied_ordering_gives_the_same_solution():
    """oq_setup_ordered (test infrastructure for KKT matrices whose exact minimum-degree ordering takes minutes): any
    elimination order yields the same iterates up to round-off; a non-permutation is rejected."""
    from osqp_solver_amd import problems as PR
    pr = PR.random_box_qp(1, n=40, mg=30, nnz_per_row=4)
    P, A = PR.qp_matrices(pr, 0)
    ref = O.OracleQPSolver(P, pr["q"][0], A, pr["l"][0], pr["u"][0])
    st0, x0 = ref.solve()
    N = 40 + 70
    rng = np.random.default_rng(5)
    for perm in (np.arange(N), np.arange(N)[::-1].copy(), rng.permutation(N)):
        o = O.OracleQPSolver(P, pr["q"][0], A, pr["l"][0], pr["u"][0], kkt_perm=perm)
        st, x = o.solve()
        assert st == st0 and o.info().iter == ref.info().iter and np.max(np.abs(x - x0)) < 1e-9
    bad = np.arange(N); bad[3] = bad[4]
    with pytest.raises(ValueError):
        O.OracleQPSolver(P, pr["q"][0], A, pr["l"][0], pr["u"][0], kkt_perm=bad)
