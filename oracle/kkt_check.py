"""oracle/kkt_check.py -- problem-intrinsic optimality check (TEST INFRASTRUCTURE ONLY).

Independent of any solver: evaluates the KKT conditions of
    min 1/2 x'Px + q'x   s.t.  l <= Ax <= u
at a candidate (x, y) with numpy/scipy only.  This is what pins the oracle in
the absence of reference outputs.
"""
import numpy as np
import scipy.sparse as sp

INF = 1e30  # [REF] /root/reference/src/constraints/constraints.h:11


def sym_from_any(P):
    """Full symmetric matrix from a matrix holding the upper triangle, or both."""
    P = sp.csc_matrix(P)
    U = sp.triu(P, 0)
    return (U + sp.triu(P, 1).T).tocsc()


def kkt_residuals(P, q, A, l, u, x, y):
    Pf = sym_from_any(P)
    A = sp.csc_matrix(A)
    q = np.zeros(A.shape[1]) if q is None else np.asarray(q, float)
    Ax = A @ x
    prim = max(0.0, float(np.max(np.maximum(l - Ax, 0.0), initial=0.0)), float(np.max(np.maximum(Ax - u, 0.0), initial=0.0)))
    stat = float(np.max(np.abs(Pf @ x + q + A.T @ y), initial=0.0))
    # complementarity: y_i>0 only at the upper bound, y_i<0 only at the lower
    yp, ym = np.maximum(y, 0.0), np.minimum(y, 0.0)
    gap_u = np.where(u < INF * 1e-4, u - Ax, 0.0)
    gap_l = np.where(l > -INF * 1e-4, Ax - l, 0.0)
    comp = float(max(np.max(np.abs(yp * gap_u), initial=0.0), np.max(np.abs(ym * gap_l), initial=0.0)))
    # multipliers on infinite bounds must vanish
    dual_sign = float(max(np.max(np.where(u >= INF * 1e-4, yp, 0.0), initial=0.0),
                          np.max(np.where(l <= -INF * 1e-4, -ym, 0.0), initial=0.0)))
    obj = float(0.5 * x @ (Pf @ x) + q @ x)
    return dict(prim=prim, stat=stat, comp=comp, dual_sign=dual_sign, obj=obj)
