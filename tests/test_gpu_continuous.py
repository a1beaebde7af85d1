"""GPU (-m gpu): continuous batching (mi_osqp.h "continuous batching": per-QP entry points + advance / poll).

The reference's SQP loop is per trajectory ([REF] /root/reference/src/gomp-solver.h:70-88): the QPs of a batch do not
finish together.  Every QP driven through solve_begin_some / advance / poll must take exactly the iterations of a
blocking solve of its own - same exit code, iteration count, rho updates, and the same solution BIT FOR BIT - whatever
the other QPs of the handle are doing; the blocking path itself is pinned against the oracle (1e-6 on x) here as well."""
import numpy as np
import pytest

import osqp_solver_amd as M
from oracle import oracle as O
from osqp_solver_amd import problems as PR

pytestmark = pytest.mark.gpu
TOL_X = 1e-6


def _make(pr, **kw):
    return M.BatchSolver(pr["P"], pr["Px"], pr["q"], pr["A"], pr["Ax"], pr["l"], pr["u"], **kw)


def _drain(s, max_advances=400):
    """advance / poll until nothing is iterating; returns the ids in the order they finished"""
    order = []
    for _ in range(max_advances):
        if not s.running():
            return order
        s.advance(1)
        order += list(s.poll(True))
    raise AssertionError("continuous solve did not finish")


def _same(info_a, x_a, info_b, x_b):
    assert (info_a.status_val, info_a.iter, info_a.rho_updates) == (info_b.status_val, info_b.iter, info_b.rho_updates)
    assert info_a.rho == info_b.rho and info_a.obj_val == info_b.obj_val and info_a.pri_res == info_b.pri_res
    assert np.array_equal(x_a, x_b, equal_nan=True)


@pytest.mark.parametrize("tile", [1, 2])
def test_staggered_begins_equal_blocking_solves_bitwise(tile, monkeypatch):
    monkeypatch.setenv("MI_OSQP_TILE", str(tile))
    B = 11
    pr = PR.random_box_qp(B, n=96, mg=64, nnz_per_row=6)
    ref = _make(pr)
    info_ref, x_ref = ref.solve(), ref.primal()
    for b in (0, 5):                                   # the blocking path against the oracle
        P, A = PR.qp_matrices(pr, b)
        o = O.OracleQPSolver(P, pr["q"][b], A, pr["l"][b], pr["u"][b])
        st, xo = o.solve()
        assert st == info_ref[b].status_val and o.info().iter == info_ref[b].iter
        assert np.max(np.abs(x_ref[b] - xo)) <= TOL_X
    assert len({i.iter for i in info_ref}) > 1         # the QPs do not finish together
    s = _make(pr)
    assert s.stats()["tile"] == tile
    first, second = [0, 2, 3, 7, 10], [1, 4, 5, 6, 8, 9]
    s.solve_begin_some(first)
    s.advance(1); done = list(s.poll(True))
    s.advance(1); done += list(s.poll(True))
    s.solve_begin_some(second)                         # these start two segments later: own iteration counts
    done += _drain(s)
    assert sorted(done) == list(range(B))
    info, x = s.info_some(range(B)), s.primal_some(range(B))
    for b in range(B):
        _same(info[b], x[b], info_ref[b], x_ref[b])
    # pipelined: two advances in flight before the first poll
    s2 = _make(pr)
    s2.solve_begin_some(range(B))
    s2.advance(1)
    got = []
    for _ in range(200):
        if not s2.running():
            break
        s2.advance(1)
        got += list(s2.poll(True))
    got += list(s2.poll(True))
    assert sorted(got) == list(range(B))
    info2, x2 = s2.info_some(range(B)), s2.primal_some(range(B))
    for b in range(B):
        _same(info2[b], x2[b], info_ref[b], x_ref[b])
    # a blocking call ends the continuous mode; the whole-batch getters see the continuous results
    assert np.array_equal(s2.primal(), x_ref)
    info3 = s2.solve()                                  # warm re-solve of everything, blocking
    ref.solve()
    for b in range(B):
        _same(info3[b], s2.primal()[b], ref.info()[b], ref.primal()[b])


def test_update_and_warm_start_some_equal_the_whole_batch_calls_bitwise():
    B = 9
    pr = PR.random_box_qp(B, n=80, mg=56, nnz_per_row=5)
    rng = np.random.default_rng(7)
    Ax2 = pr["Ax"] * (1.0 + 0.2 * rng.standard_normal(pr["Ax"].shape))
    l2, u2 = pr["l"] * 1.3, pr["u"] * 0.8
    xw = 0.1 * rng.standard_normal((B, 80))
    ref = _make(pr)
    ref.solve()
    ref.update_A_bounds(Ax2, l2, u2)
    ref.warm_start_x(xw)
    info_ref, x_ref = ref.solve(), ref.primal()
    s = _make(pr)
    s.solve_begin_some(range(B))
    _drain(s)
    # the update reaches the QPs in two groups, the second one while the first is already iterating again
    g1, g2 = [1, 4, 6, 8], [0, 2, 3, 5, 7]
    s.update_A_bounds_some(g1, Ax2[g1], l2[g1], u2[g1])
    s.warm_start_x_some(g1, xw[g1])
    s.solve_begin_some(g1)
    s.advance(1); fin = list(s.poll(True))
    s.update_A_bounds_some(g2, Ax2[g2], l2[g2], u2[g2])
    s.warm_start_x_some(g2, xw[g2])
    s.solve_begin_some(g2)
    fin += _drain(s)
    assert sorted(fin) == list(range(B))
    info, x = s.info_some(range(B)), s.primal_some(range(B))
    for b in range(B):
        _same(info[b], x[b], info_ref[b], x_ref[b])
    for b in (0, 4):
        P, A = PR.qp_matrices(pr, b)
        o = O.OracleQPSolver(P, pr["q"][b], A, pr["l"][b], pr["u"][b])
        o.solve()
        A2 = A.copy(); A2.data[:] = Ax2[b]
        o.update(l2[b], A2, u2[b]); o.set_warm_start(xw[b])
        st, xo = o.solve()
        assert st == info[b].status_val and o.info().iter == info[b].iter and np.max(np.abs(x[b] - xo)) <= TOL_X


@pytest.mark.parametrize("tail", ["", "64"])
def test_reinit_some_equals_a_fresh_setup_bitwise(tail, monkeypatch):
    if tail:
        monkeypatch.setenv("MI_OSQP_DENSE_TAIL", tail)            # the refactorisation list with the dense-tail kernels
    B = 7
    pr = PR.random_box_qp(B, n=128, mg=96, nnz_per_row=6)
    rng = np.random.default_rng(11)
    Ax2 = pr["Ax"] * (1.0 + 0.3 * rng.standard_normal(pr["Ax"].shape))
    l2, u2 = pr["l"] * 0.7, pr["u"] * 1.1
    xw = 0.05 * rng.standard_normal((B, 128))
    pr2 = dict(pr, Ax=Ax2, l=l2, u=u2)
    fresh = _make(pr2)
    if tail:
        assert fresh.stats()["dense_tail_rows"] == 64
    fresh.warm_start_x(xw)
    info_ref, x_ref = fresh.solve(), fresh.primal()
    s = _make(pr)                                      # built with other A values and bounds, solved (rho adapted), then re-initialised
    s.solve_begin_some(range(B))
    _drain(s)
    ids = [6, 0, 3, 1, 5, 2, 4]
    s.reinit_some(ids, Ax2[ids], l2[ids], u2[ids])
    s.warm_start_x_some(ids, xw[ids])
    s.solve_begin_some(ids)
    _drain(s)
    info, x = s.info_some(range(B)), s.primal_some(range(B))
    for b in range(B):
        _same(info[b], x[b], info_ref[b], x_ref[b])
    assert max(i.rho_updates for i in info) >= 1       # the device-built refactorisation list was used
    # reset() restores the snapshot the per-QP calls kept up to date
    s.reset()
    s.warm_start_x(xw)
    info4 = s.solve()
    for b in range(B):
        _same(info4[b], s.primal()[b], info_ref[b], x_ref[b])


def test_failure_isolation_and_errors_in_continuous_mode():
    B = 4
    pr = PR.random_box_qp(B, n=48, mg=32, nnz_per_row=4)
    Px = pr["Px"].copy()
    Px[2] = -np.abs(Px[2]) - 1.0                       # QP 2: indefinite P -> no valid factor
    s = M.BatchSolver(pr["P"], Px, pr["q"], pr["A"], pr["Ax"], pr["l"], pr["u"])
    s.solve_begin_some(range(B))
    with pytest.raises(M.MiOsqpError):
        s.warm_start_x_some([1], np.zeros(48))         # QP 1 is iterating
    fin = _drain(s)
    assert sorted(fin) == list(range(B))
    info = s.info_some(range(B))
    assert info[2].status_val == -7 and np.all(np.isnan(s.primal_some([2])))
    assert all(info[b].status_val == 1 for b in (0, 1, 3))
    ref = M.BatchSolver(pr["P"], Px, pr["q"], pr["A"], pr["Ax"], pr["l"], pr["u"])
    iref = ref.solve()
    for b in (0, 1, 3):
        _same(info[b], s.primal_some([b])[0], iref[b], ref.primal()[b])
    with pytest.raises(M.MiOsqpError):
        s.solve_begin_some([B])


def test_blocking_solve_through_the_advance_kernel_is_bitwise_the_segment_loop(monkeypatch):
    """MI_OSQP_ADVANCE_SOLVE=1 (opt-in): a batch of no more tiles than CUs solved by advance_kernel launches - every QP runs
    to its end or to its next rho update inside one launch, the host only refactors the paused QPs - instead of two launches
    and a synchronisation per segment.  Same results bit for bit, also when a QP runs into max_iter on a rho-update iteration
    (it finishes AND is refactored: the warm re-solve continues from that factor)."""
    B = 9
    pr = PR.random_box_qp(B, n=96, mg=64, nnz_per_row=6)
    for kw in ({}, dict(max_iter=100, check_termination=0), dict(max_iter=50, check_termination=30, eps_abs=1e-9, eps_rel=1e-9)):
        ref = _make(pr, **kw)
        i1, x1 = ref.solve(), ref.primal().copy()
        i2, x2 = ref.solve(), ref.primal().copy()
        monkeypatch.setenv("MI_OSQP_ADVANCE_SOLVE", "1")
        s = _make(pr, **kw)
        j1, y1 = s.solve(), s.primal().copy()
        j2, y2 = s.solve(), s.primal().copy()
        monkeypatch.delenv("MI_OSQP_ADVANCE_SOLVE")
        for b in range(B):
            _same(j1[b], y1[b], i1[b], x1[b])
            _same(j2[b], y2[b], i2[b], x2[b])
        assert s.last_solve_stats()["launches"] <= ref.last_solve_stats()["launches"]


def test_a_staging_ring_that_wraps_every_few_calls_changes_nothing(monkeypatch):
    """The per-QP calls stage ids, rows, scaled values and warm starts in a pinned ring; a call's spans come from one lap
    (solver.hip ring_reserve).  With a ring barely larger than one whole-batch call (MI_OSQP_CONT_RING_KB) every second call
    wraps while the previous calls' transfers and kernels are still in flight: many rounds of partial updates / warm starts /
    solves must stay bitwise equal to the blocking calls."""
    B, n = 12, 80
    pr = PR.random_box_qp(B, n=n, mg=56, nnz_per_row=5)
    rng = np.random.default_rng(11)
    ref = _make(pr)
    ref.solve()
    monkeypatch.setenv("MI_OSQP_CONT_RING_KB", "1")
    s = _make(pr)
    s.solve_begin_some(range(B)); _drain(s)
    for rnd in range(12):
        Ax2 = pr["Ax"] * (1.0 + 0.1 * rng.standard_normal(pr["Ax"].shape))
        l2, u2 = pr["l"] * (1.0 + 0.05 * rnd), pr["u"] * (1.0 - 0.02 * rnd)
        xw = 0.1 * rng.standard_normal((B, n))
        ref.update_A_bounds(Ax2, l2, u2); ref.warm_start_x(xw)
        info_ref, x_ref = ref.solve(), ref.primal()
        order = rng.permutation(B)
        groups = [sorted(order[:5].tolist()), sorted(order[5:9].tolist()), sorted(order[9:].tolist())]
        fin = []
        for k, g in enumerate(groups):                      # at most one poll behind: the groups' stagings queue up in the ring
            s.update_A_bounds_some(g, Ax2[g], l2[g], u2[g])
            s.warm_start_x_some(g, xw[g])
            s.solve_begin_some(g)
            if k >= 2:
                fin += list(s.poll(True))
            s.advance(1)
        fin += list(s.poll(True)); fin += list(s.poll(True))
        fin += _drain(s)
        assert sorted(fin) == list(range(B))
        info, x = s.info_some(range(B)), s.primal_some(range(B))
        for b in range(B):
            _same(info[b], x[b], info_ref[b], x_ref[b])
