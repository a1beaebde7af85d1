"""Synthetic workloads of BASELINE.json `configs` (inputs of the hot path).

* random_box_qp  -- config 3: batch of box-QPs, ONE shared pattern, per-QP values
                    (generator frozen in SURVEY.md section 8(d)).
* GompBuilder    -- joint-space part of the reference's ConstraintBuilder
                    ([REF] /root/reference/src/constraints/constraint-builder.h:30-88,
                    124-151,185-219) restated with numpy/scipy; pinned by the
                    reference's own known-answer tests ([REF] tests/test.cpp:45-248,
                    fixtures in tests/golden/constraint_builder_kats.json).
* gomp_qp        -- configs 2 and 4: the QP GOMPSolver::run builds for one
                    trajectory with no robot balls / obstacles
                    ([REF] src/gomp-solver.h:18-36,57-64,118-139; src/utils.h:50-64).

Pure data generation: nothing here solves anything.
"""
import numpy as np
import scipy.sparse as sp

INF = 1e30   # [REF] src/constraints/constraints.h:11


# ----------------------------------------------------------------- config 3

def random_box_qp(B, n=512, mg=512, nnz_per_row=8, half_bw=4, pattern_seed=1234, value_seed=1000):
    """A = [I_n ; G] (m = n + mg), G has `nnz_per_row` entries per row at uniformly
    random columns (pattern shared by the whole batch); P = diag(d) + S with S a
    symmetric band of half-bandwidth `half_bw`, made strictly diagonally dominant;
    q ~ N(0,1); l = -U(0.1,1), u = +U(0.1,1).  Returns a dict with scipy CSC
    patterns (data = pattern ones) and per-QP value arrays in CSC order."""
    rng = np.random.default_rng(pattern_seed)
    rows, cols = [], []
    for r in range(mg):
        c = rng.choice(n, size=nnz_per_row, replace=False)
        rows += [n + r] * nnz_per_row
        cols += list(c)
    rows = list(range(n)) + rows
    cols = list(range(n)) + cols
    m = n + mg
    A_pat = sp.csc_matrix((np.ones(len(rows)), (rows, cols)), shape=(m, n))
    A_pat.sort_indices()
    # band pattern of P (upper triangle only is passed; both are legal)
    pr, pc = [], []
    for i in range(n):
        for k in range(0, half_bw + 1):
            if i + k < n:
                pr.append(i); pc.append(i + k)
    P_pat = sp.csc_matrix((np.ones(len(pr)), (pr, pc)), shape=(n, n))
    P_pat.sort_indices()
    nnzA, nnzP = A_pat.nnz, P_pat.nnz
    # CSC order bookkeeping
    Acoo = A_pat.tocoo()
    # entries of the identity block have row < n
    A_is_ident = (A_pat.indices < n)
    P_rows = P_pat.indices
    P_cols = np.repeat(np.arange(n), np.diff(P_pat.indptr))
    Px = np.empty((B, nnzP)); Ax = np.empty((B, nnzA)); q = np.empty((B, n))
    l = np.empty((B, m)); u = np.empty((B, m))
    for b in range(B):
        r = np.random.default_rng(value_seed + b)
        av = r.standard_normal(nnzA)
        av[A_is_ident] = 1.0
        Ax[b] = av
        d = r.uniform(0.5, 1.5, n)
        off = 0.1 * r.standard_normal(nnzP)
        offd = P_rows != P_cols
        # row sums of |S| for diagonal dominance
        rs = np.zeros(n)
        np.add.at(rs, P_rows[offd], np.abs(off[offd]))
        np.add.at(rs, P_cols[offd], np.abs(off[offd]))
        pv = np.where(offd, off, 0.0)
        pv[~offd] = d + rs
        Px[b] = pv
        q[b] = r.standard_normal(n)
        l[b] = -r.uniform(0.1, 1.0, m)
        u[b] = r.uniform(0.1, 1.0, m)
    del Acoo
    return dict(n=n, m=m, P=P_pat, A=A_pat, Px=Px, Ax=Ax, q=q, l=l, u=u)


def grid_qp(g, seed=7):
    """Config-5 style structured sparse QP at a size that fits the LDS-resident solver:
    variables on a g x g grid, P = 5-point Laplacian + I (strictly convex),
    A = [I ; Dx ; Dy] (boxes on the variables and on their first differences),
    so n = g^2, m = n + 2 g (g - 1).  The KKT factor has the deep, nested-dissection-like
    elimination tree of a 2-D mesh.  Returns a batch dict with B = 1."""
    r = np.random.default_rng(seed)
    n = g * g
    idx = np.arange(n).reshape(g, g)
    rows, cols, vals = [], [], []
    def add(i, j, v):
        rows.append(i); cols.append(j); vals.append(v)
    for a in range(g):
        for b in range(g):
            add(idx[a, b], idx[a, b], 5.0)
            if a + 1 < g: add(min(idx[a, b], idx[a + 1, b]), max(idx[a, b], idx[a + 1, b]), -1.0)
            if b + 1 < g: add(min(idx[a, b], idx[a, b + 1]), max(idx[a, b], idx[a, b + 1]), -1.0)
    P = sp.csc_matrix((vals, (rows, cols)), shape=(n, n)); P.sort_indices()
    ar, ac, av = list(range(n)), list(range(n)), [1.0] * n
    k = n
    for a in range(g):
        for b in range(g - 1):
            ar += [k, k]; ac += [idx[a, b + 1], idx[a, b]]; av += [1.0, -1.0]; k += 1
    for a in range(g - 1):
        for b in range(g):
            ar += [k, k]; ac += [idx[a + 1, b], idx[a, b]]; av += [1.0, -1.0]; k += 1
    m = k
    A = sp.csc_matrix((av, (ar, ac)), shape=(m, n)); A.sort_indices()
    q = r.standard_normal(n) * 3.0
    l = np.concatenate([-r.uniform(0.2, 1.0, n), -r.uniform(0.05, 0.3, m - n)])
    u = np.concatenate([r.uniform(0.2, 1.0, n), r.uniform(0.05, 0.3, m - n)])
    return dict(n=n, m=m, P=P, A=A, Px=P.data[None].copy(), Ax=A.data[None].copy(), q=q[None], l=l[None], u=u[None])


def qp_matrices(prob, b):
    """scipy matrices (P upper triangle, A) of QP b of a batch dict."""
    P = prob["P"].copy(); P.data = prob["Px"][b].copy()
    A = prob["A"].copy(); A.data = prob["Ax"][b].copy()
    return P, A


# ------------------------------------------------------------ GOMP builder

def tri_diagonal_matrix(a, b, n, offset=0, diagonal_num=1):
    """[REF] src/utils.h:50-64 -- both triangles are emitted, as the reference does."""
    r, c, v = [], [], []
    for i in range(offset, n):
        r.append(i); c.append(i); v.append(a)
        if i + diagonal_num < n:
            r.append(i); c.append(i + diagonal_num); v.append(b)
        if i - diagonal_num >= offset:
            r.append(i); c.append(i - diagonal_num); v.append(b)
    return sp.csc_matrix((v, (r, c)), shape=(n, n))


class GompBuilder:
    """Joint-space rows of ConstraintBuilder<N_DIM> (no balls, `n_obstacle_rows`
    extra all-zero rows per waypoint exactly as the reference over-allocates them,
    [REF] constraint-builder.h:43-44).  A constraint is (low, upp) with None = absent."""

    def __init__(self, dims, waypoints, n_obstacles=0, n_balls=0):
        self.D, self.W = dims, waypoints
        self.lower, self.upper = [], []
        self.triplets = {}            # (row, col) -> value, last write wins (:129)
        self._link_velocity_to_position()
        self.offset = len(self.lower)
        extra = dims * (waypoints + waypoints - 1 + waypoints - 2 + waypoints * (3 + n_obstacles * n_balls))
        self.lower += [-INF] * extra
        self.upper += [INF] * extra

    # index helpers ([REF] :138-151)
    def nth_pos(self, i):
        return i * self.D

    def nth_velocity(self, i):
        return self.W * self.D + i * self.D

    def nth_acceleration(self, i):
        return self.W * self.D + (self.W - 1) * self.D + i * self.D

    def _add(self, row, eq, lo, up):
        for col, coeff in eq:
            self.triplets[(row, col)] = coeff
        if lo is not None:
            self.lower[row] = lo
        if up is not None:
            self.upper[row] = up
        assert self.lower[row] <= self.upper[row]

    @staticmethod
    def _nth(c, j):
        lo, up = c
        return (None if lo is None else float(lo[j])), (None if up is None else float(up[j]))

    def _link_velocity_to_position(self):      # [REF] :203-219
        for i in range(self.W - 1):
            bv, bp, bn = self.nth_velocity(i), self.nth_pos(i), self.nth_pos(i + 1)
            for j in range(self.D):
                self.lower.append(-INF); self.upper.append(INF)
                self._add(len(self.lower) - 1, [(bv + j, 1.0), (bn + j, -1.0), (bp + j, 1.0)], 0.0, 0.0)

    def _variables_in_range(self, first, last, c):   # [REF] :185-201
        for start in range(first, last + 1, self.D):
            for j in range(self.D):
                lo, up = self._nth(c, j)
                self._add(self.offset + start + j, [(start + j, 1.0)], lo, up)
        return self

    def position(self, i, c):
        return self.positions(i, i, c)

    def positions(self, first, last, c):
        return self._variables_in_range(self.nth_pos(first), self.nth_pos(last), c)

    def velocity(self, i, c):
        return self.velocities(i, i, c)

    def velocities(self, first, last, c):
        assert first <= last < self.W - 1
        return self._variables_in_range(self.nth_velocity(first), self.nth_velocity(last), c)

    def acceleration(self, i, c):                    # [REF] :72-88
        assert i + 2 < self.W
        ba, bv, bn = self.nth_acceleration(i), self.nth_velocity(i), self.nth_velocity(i + 1)
        for j in range(self.D):
            lo, up = self._nth(c, j)
            self._add(self.offset + ba + j, [(bn + j, 1.0), (bv + j, -1.0)], lo, up)
        return self

    def accelerations(self, first, last, c):
        for i in range(first, last + 1):
            self.acceleration(i, c)
        return self

    def build(self):                                 # [REF] :124-136
        m, n = len(self.lower), 2 * self.D * self.W
        if self.triplets:
            rc = np.array(list(self.triplets.keys()))
            v = np.array(list(self.triplets.values()), float)
            A = sp.csc_matrix((v, (rc[:, 0], rc[:, 1])), shape=(m, n))
        else:
            A = sp.csc_matrix((m, n))
        A.sort_indices()
        return np.array(self.lower), A, np.array(self.upper)


def equal(v):
    v = np.asarray(v, float)
    return (v, v)


def in_range(lo, up):
    return (None if lo is None else np.asarray(lo, float), None if up is None else np.asarray(up, float))


def scaled(c, f):
    lo, up = c
    return (None if lo is None else lo * f, None if up is None else up * f)


def gomp_qp(dims, waypoints, start, end, time_step=0.1,
            q_lim=2 * np.pi, v_lim=np.pi, a_lim=np.pi * 800 / 180):
    """(P, (l, A, u), warm_start) of GOMPSolver::run's QP for one segment without
    balls/obstacles; limits default to [REF] examples/solver-example.cpp:44-46."""
    D, W = dims, waypoints
    assert W >= 4
    pos_con = in_range(np.full(D, -q_lim), np.full(D, q_lim))
    vel_con = scaled(in_range(np.full(D, -v_lim), np.full(D, v_lim)), time_step)            # [REF] gomp-solver.h:29
    acc_con = scaled(in_range(np.full(D, -a_lim), np.full(D, a_lim)), time_step * time_step)  # :30
    zero = equal(np.zeros(D))
    b = GompBuilder(D, W)
    (b.position(0, equal(start)).positions(1, W - 2, pos_con).position(W - 3, equal(end))     # :130-133
      .velocities(0, W - 4, vel_con).velocity(W - 3, zero)                                     # :134-135
      .accelerations(0, W - 4, acc_con).acceleration(W - 3, zero))                             # :136-137
    l, A, u = b.build()
    P = tri_diagonal_matrix(2.0, -1.0, 2 * D * W, D * W, D)                                    # :63
    # warm start of [REF] gomp-solver.h:105-116: linspace positions, zero velocities
    pos = np.linspace(np.asarray(start, float), np.asarray(end, float), W).reshape(-1)
    warm = np.concatenate([pos, np.zeros(D * W)])
    return P, (l, A, u), warm


def gomp_batch(B, dims, waypoints, seed=2000, **kw):
    """B joint-space GOMP QPs (random start/end ~ U(-pi,pi)^D, seed 2000+b): one
    shared pattern, identical matrix values, per-trajectory bounds."""
    P0 = A0 = None
    ls, us, warms = [], [], []
    for b in range(B):
        r = np.random.default_rng(seed + b)
        s, e = r.uniform(-np.pi, np.pi, dims), r.uniform(-np.pi, np.pi, dims)
        P, (l, A, u), w = gomp_qp(dims, waypoints, s, e, **kw)
        if P0 is None:
            P0, A0 = P, A
        ls.append(l); us.append(u); warms.append(w)
    n, m = A0.shape[1], A0.shape[0]
    P0 = sp.csc_matrix(P0); P0.sort_indices()
    return dict(n=n, m=m, P=P0, A=A0, Px=np.tile(P0.data, (B, 1)), Ax=np.tile(A0.data, (B, 1)), q=None,
                l=np.array(ls), u=np.array(us), warm=np.array(warms))
