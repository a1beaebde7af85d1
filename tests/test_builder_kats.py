"""CPU: the GOMP QP generator (osqp-solver_amd/problems.py) against the
reference's own known-answer tests ([REF] tests/test.cpp:45-248), stored as data
in tests/golden/constraint_builder_kats.json."""
import numpy as np

from osqp_solver_amd import problems as PR


def _offsets(W, D):
    fp = (W - 1) * D
    fv = fp + W * D
    fa = fv + (W - 1) * D
    return {"firstPosition": fp, "firstVelocity": fv, "firstAcceleration": fa, "first3dPosition": fa + (W - 2) * D}


def _build(k):
    b = PR.GompBuilder(k["dims"], k["waypoints"])
    for name, first, last, lo, up in k.get("calls", []):
        getattr(b, name)(first, last, PR.in_range(lo, up))
    return b.build()


def test_indices(builder_kats):
    ix = builder_kats["indices"]
    b = PR.GompBuilder(ix["position"]["dims"], ix["position"]["waypoints"])
    for i, v in ix["position"]["nthPos"].items():
        assert b.nth_pos(int(i)) == v
    for i, v in ix["velocity"]["nthVelocity"].items():
        assert b.nth_velocity(int(i)) == v
    b = PR.GompBuilder(ix["acceleration"]["dims"], ix["acceleration"]["waypoints"])
    for i, v in ix["acceleration"]["nthAcceleration"].items():
        assert b.nth_acceleration(int(i)) == v


def test_known_answer_blocks(builder_kats):
    for name in ("linkingVelocityToPosition", "jointPosition", "velocity", "acceleration", "all"):
        k = builder_kats[name]
        l, A, u = _build(k)
        D, W = k["dims"], k["waypoints"]
        assert A.shape == ((W - 1) * D + D * (W + W - 1 + W - 2 + 3 * W), 2 * D * W)    # [REF] constraint-builder.h:43-44
        sr = k["start_row"] if isinstance(k["start_row"], int) else _offsets(W, D)[k["start_row"]]
        rows = len(k["l"])
        np.testing.assert_array_equal(A.toarray()[sr:sr + rows], np.array(k["A"], float), err_msg=name)
        np.testing.assert_array_equal(l[sr:sr + rows], k["l"], err_msg=name)
        np.testing.assert_array_equal(u[sr:sr + rows], k["u"], err_msg=name)


def test_gomp_qp_shapes_match_survey_table():
    # config 2 (D=6,W=50) and config 4 (D=7,W=100)
    for D, W, n, m, nnzA, nnzPtriu in ((6, 50, 600, 2076, 2040, 594), (7, 100, 1400, 4872, 4830, 1393)):
        P, (l, A, u), warm = PR.gomp_qp(D, W, np.zeros(D), np.ones(D))
        assert A.shape == (m, n) and A.nnz == nnzA and len(warm) == n
        import scipy.sparse as sp
        assert sp.triu(P).nnz == nnzPtriu
        assert np.all(l <= u)
        # goal pinned at waypoint W-3, start at waypoint 0 ([REF] gomp-solver.h:130-133)
        off = (W - 1) * D
        assert np.all(l[off:off + D] == 0) and np.all(u[off + (W - 3) * D: off + (W - 2) * D] == 1)


def test_tri_diagonal_matrix():
    M = PR.tri_diagonal_matrix(2, -1, 8, 4, 2).toarray()          # [REF] src/utils.h:50-64
    assert np.all(M[:4] == 0) and np.all(M[:, :4] == 0)
    np.testing.assert_array_equal(M[4:, 4:], [[2, 0, -1, 0], [0, 2, 0, -1], [-1, 0, 2, 0], [0, -1, 0, 2]])
