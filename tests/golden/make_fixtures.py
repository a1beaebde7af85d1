"""Generates tests/golden/qp_fixtures.json (run in the build container only).

Every fixture = inputs (dense P upper/both triangles, q, A, l, u) + expected
outputs.  Where a closed form exists the expectation comes from numpy alone
(`source` = "closed_form"); otherwise from the CPU oracle run to eps=1e-9 and
accepted only if the solver-independent KKT check (oracle/kkt_check.py) passes
at 1e-6 (`source` = "oracle+kkt").  The reference itself cannot generate
vectors: its solver is an un-vendored network dependency (Oracle).
"""
import json
import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O                      # noqa: E402
from oracle.kkt_check import kkt_residuals          # noqa: E402
import osqp_solver_amd.problems as PR               # noqa: E402

INF = 1e30
out = []


def add(name, P, q, A, l, u, x=None, y=None, status="kOptimal", source="oracle+kkt", note=""):
    P = np.asarray(sp.csc_matrix(P).todense()); A = np.asarray(sp.csc_matrix(A).todense())
    if x is None and status == "kOptimal":
        s = O.OracleQPSolver(P, q, A, l, u, eps_abs=1e-9, eps_rel=1e-9, max_iter=200000)
        st, x = s.solve(); y = s.y
        assert st == 1, (name, st)
        r = kkt_residuals(P, q, A, np.asarray(l, float), np.asarray(u, float), x, y)
        assert max(r["prim"], r["stat"], r["comp"], r["dual_sign"]) < 1e-6, (name, r)
    out.append(dict(name=name, n=int(A.shape[1]), m=int(A.shape[0]), P=P.tolist(),
                    q=None if q is None else list(map(float, q)), A=A.tolist(),
                    l=list(map(float, l)), u=list(map(float, u)),
                    x=None if x is None else list(map(float, x)), y=None if y is None else list(map(float, y)),
                    status=status, source=source, note=note))


# 1. the QP of upstream's "setup and solve" documentation page
add("osqp_doc_demo", [[4, 1], [1, 2]], [1, 1], [[1, 1], [1, 0], [0, 1]], [1, 0, 0], [1, 0.7, 0.7],
    x=[0.3, 0.7], y=[-2.9, 0.0, 0.2], source="closed_form",
    note="active set {row0 eq, row2 upper}: solve the 3x3 KKT system by hand -> x=(0.3,0.7), y=(-2.9,0,0.2), obj 1.88")

# 2. box-constrained diagonal QP: x_i = clip(-q_i/p_i, l_i, u_i)
rng = np.random.default_rng(7)
n = 12
p = rng.uniform(0.5, 3.0, n); q = rng.standard_normal(n) * 3
l = -rng.uniform(0.1, 1.0, n); u = rng.uniform(0.1, 1.0, n)
x = np.clip(-q / p, l, u)
y = -(p * x + q)
add("diag_box", np.diag(p), q, np.eye(n), l, u, x=x, y=y, source="closed_form")

# 3. equality-constrained QP: one linear solve
n, me = 10, 4
M = rng.standard_normal((n, n)); P = M @ M.T + np.eye(n)
A = rng.standard_normal((me, n)); b = rng.standard_normal(me); q = rng.standard_normal(n)
K = np.block([[P, A.T], [A, np.zeros((me, me))]])
sol = np.linalg.solve(K, np.concatenate([-q, b]))
add("equality_only", np.triu(P), q, A, b, b, x=sol[:n], y=sol[n:], source="closed_form")

# 4. generic small QPs (oracle + KKT check)
for k, (n, m) in enumerate([(5, 8), (15, 25), (30, 40)]):
    r = np.random.default_rng(100 + k)
    M = sp.random(n, n, density=0.3, random_state=100 + k).toarray()
    P = M @ M.T + 0.1 * np.eye(n)
    A = sp.random(m, n, density=0.4, random_state=200 + k).toarray()
    A[:n, :] += np.eye(n)[: min(m, n), :][:n] if m >= n else 0
    q = r.standard_normal(n)
    l = -r.uniform(0.2, 1.5, m); u = r.uniform(0.2, 1.5, m)
    l[::5] = -INF; u[3::7] = INF
    add(f"generic_{n}x{m}", np.triu(P), q, A, l, u)

# 5. infeasible / unbounded
add("primal_infeasible", [[1, 0], [0, 1]], [1, 1], [[1, 0], [1, 0], [0, 1]], [1, -INF, -1], [INF, 0, 1],
    status="kPrimalInfeasible", source="by_construction", note="x0 >= 1 and x0 <= 0")
add("dual_infeasible", [[0, 0], [0, 1]], [-1, 0], [[1, 0], [0, 1]], [0, -1], [INF, 1],
    status="kDualInfeasible", source="by_construction", note="min -x0 with x0 >= 0 only: unbounded below")

# 6. a tiny GOMP segment (D=2, W=6) in the reference's own formulation
P, (l, A, u), warm = PR.gomp_qp(2, 6, [0.0, 0.1], [0.5, -0.3])
add("gomp_2x6", P, None, A, l, u, note="P holds both triangles, as the reference passes it")

path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "qp_fixtures.json")
with open(path, "w") as f:
    json.dump(out, f, indent=0)
print("wrote", path, len(out), "fixtures")
