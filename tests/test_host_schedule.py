"""CPU: host logic of the product -- ordering, symbolic + left-looking LDL',
Ruiz scaling, and the DEVICE schedules replayed sequentially on the host
(mi_osqp_debug_host_kkt_solve) -- against the oracle's independent factor."""
import numpy as np
import pytest
import scipy.sparse as sp

import osqp_solver_amd as M
from oracle import oracle as O
from osqp_solver_amd import problems as PR


def _cases():
    pr = PR.random_box_qp(1, n=64, mg=48, nnz_per_row=4)
    P, A = PR.qp_matrices(pr, 0)
    yield "box64", P, A, pr["l"][0], pr["u"][0]
    pr = PR.random_box_qp(1)
    P, A = PR.qp_matrices(pr, 0)
    yield "config3", P, A, pr["l"][0], pr["u"][0]
    P, (l, A, u), _ = PR.gomp_qp(6, 50, np.zeros(6), np.ones(6))
    yield "gomp6x50", P, A, l, u
    P, (l, A, u), _ = PR.gomp_qp(3, 5, np.zeros(3), np.ones(3))
    yield "gomp3x5", P, A, l, u
    # degenerate shapes: one variable / one row; diagonal KKT; empty P
    yield "1x1", sp.csc_matrix([[2.0]]), sp.csc_matrix([[1.0]]), np.array([-1.0]), np.array([1.0])
    yield "diag", sp.eye(5).tocsc(), sp.eye(5).tocsc(), -np.ones(5), np.ones(5)
    yield "emptyP", sp.csc_matrix((4, 4)), sp.csc_matrix(np.vstack([np.eye(4), np.ones((1, 4))])), -np.ones(5), np.ones(5)
    # dense-ish: one big supernode, exercises multi-chunk phase B
    rng = np.random.default_rng(3)
    G = rng.standard_normal((40, 40)); Pd = sp.csc_matrix(np.triu(G @ G.T + np.eye(40)))
    yield "dense40", Pd, sp.csc_matrix(rng.standard_normal((30, 40))), -np.ones(30), np.ones(30)


@pytest.mark.parametrize("case", list(_cases()), ids=lambda c: c[0])
@pytest.mark.parametrize("scaling", [10, 0])
@pytest.mark.parametrize("relax", ["0", "16"])
def test_schedule_replay_matches_direct_and_oracle(case, scaling, relax, monkeypatch):
    # relax: supernodes with explicit zeros (MI_OSQP_RELAX: 0 never, 16 always; analyze() decides by itself otherwise) - the
    # solutions are the same, the stream-size properties below belong to the fundamental supernodes
    monkeypatch.setenv("MI_OSQP_RELAX", relax)
    name, P, A, l, u = case
    n, m = A.shape[1], A.shape[0]
    rhs = np.random.default_rng(5).standard_normal(n + m)
    s_sched, s_direct, st = M.debug_host_kkt_solve(P, A, l, u, rhs, scaling=scaling)
    scale = np.max(np.abs(s_direct))
    assert np.max(np.abs(s_sched - s_direct)) <= 1e-8 * scale, name
    o = O.OracleQPSolver(P, None, A, l, u, scaling=scaling)
    ref = o.kkt_solve(rhs)
    assert np.max(np.abs(ref - s_direct)) <= 1e-8 * np.max(np.abs(ref)), name
    assert st["nnz_L"] >= st["nnz_KKT"] - st["N"] and st["fwd_levels"] >= 1
    if relax != "0":
        return
    if st["dense_tail_rows"] == 0:
        assert st["fwd_slots"] >= st["nnz_L"] and st["bwd_slots"] >= st["nnz_L"]
    else:       # the trailing triangle of L is replaced by the inverted Schur complement, streamed once
        k = st["dense_tail_rows"]
        assert k % 64 == 0 and st["dense_tail_slots"] == k * k // 2
        assert st["fwd_slots"] + st["bwd_slots"] + st["dense_tail_slots"] < 2 * st["nnz_L"]


@pytest.mark.parametrize("case", list(_cases()), ids=lambda c: c[0])
@pytest.mark.parametrize("waves", [1, 3, 16, 256])
def test_dataflow_form_of_the_sweeps_replays(case, waves):
    """The barrier-free schedule form of the large single QPs (host_core.hpp Analysis::df): the host replay advances
    the waves round robin, a step waits while one of its gathers is still to be written; it fails on a deadlock, on an
    entry written twice, on an armed entry nobody writes and on a subtracting flush whose old value the sweep writes."""
    name, P, A, l, u = case
    n, m = A.shape[1], A.shape[0]
    rhs = np.random.default_rng(6).standard_normal(n + m)
    s_sched, s_direct, st = M.debug_host_kkt_solve(P, A, l, u, rhs, tri_waves=waves)
    assert np.max(np.abs(s_sched - s_direct)) <= 1e-8 * np.max(np.abs(s_direct)), name
    assert st["dense_tail_rows"] == 0


def test_dataflow_form_on_a_grid_with_deep_elimination_tree():
    pr = PR.grid_qp(40)
    P, A = PR.qp_matrices(pr, 0)
    rhs = np.random.default_rng(7).standard_normal(pr["n"] + pr["m"])
    s_sched, s_direct, st = M.debug_host_kkt_solve(P, A, pr["l"][0], pr["u"][0], rhs, tri_waves=64)
    assert np.max(np.abs(s_sched - s_direct)) <= 1e-9 * np.max(np.abs(s_direct))
    s8, _, st8 = M.debug_host_kkt_solve(P, A, pr["l"][0], pr["u"][0], rhs)
    assert st["fwd_levels"] >= 20 and np.max(np.abs(s8 - s_sched)) <= 1e-9 * np.max(np.abs(s_direct))


def test_nonconvex_is_refused():
    P = sp.csc_matrix(-np.eye(3)); A = sp.eye(3).tocsc()
    with pytest.raises(M.MiOsqpError) as e:
        M.debug_host_kkt_solve(P, A, -np.ones(3), np.ones(3), np.ones(6))
    assert e.value.code == 4


@pytest.mark.parametrize("case", list(_cases()), ids=lambda c: c[0])
@pytest.mark.parametrize("relax", ["0", "16"])
def test_block_factor_replay_matches_left_looking(case, relax, monkeypatch):
    """The DEVICE refactorisation tables (BlockFactor), interpreted on the host,
    reproduce the host left-looking LDL' (row E13 vs E5) - with fundamental and with relaxed supernodes."""
    monkeypatch.setenv("MI_OSQP_RELAX", relax)
    name, P, A, l, u = case
    dL, dD, cnt = M.debug_host_block_factor(P, A, l, u)
    assert dL <= 1e-10 and dD <= 1e-9, (name, dL, dD)
    assert cnt["blocks"] >= 1 and cnt["storage"] >= 1


# ---- dense tail (host_core.hpp DenseTail): the trailing rows served by the inverted Schur complement

def _kkt_check(P, A, l, u, scaling=10):
    n, m = A.shape[1], A.shape[0]
    rhs = np.random.default_rng(9).standard_normal(n + m)
    s_sched, s_direct, st = M.debug_host_kkt_solve(P, A, l, u, rhs, scaling=scaling)
    assert np.max(np.abs(s_sched - s_direct)) <= 1e-8 * np.max(np.abs(s_direct))
    return st


def test_dense_tail_is_chosen_for_config3_and_can_be_switched_off(monkeypatch):
    pr = PR.random_box_qp(1)
    P, A = PR.qp_matrices(pr, 0)
    st = _kkt_check(P, A, pr["l"][0], pr["u"][0])
    assert st["dense_tail_rows"] >= 256 and st["dense_tail_rows"] % 64 == 0
    monkeypatch.setenv("MI_OSQP_DENSE_TAIL", "0")
    st0 = _kkt_check(P, A, pr["l"][0], pr["u"][0])
    assert st0["dense_tail_rows"] == 0 and st0["fwd_levels"] > 3 * st["fwd_levels"]
    # the whole point: fewer values streamed per solve
    assert st["fwd_slots"] + st["bwd_slots"] + st["dense_tail_slots"] < 0.7 * (st0["fwd_slots"] + st0["bwd_slots"])


@pytest.mark.parametrize("k", [64, 128, 192])
def test_forced_dense_tail_on_patterns_that_would_not_pick_one(k, monkeypatch):
    """Forcing a tail exercises Schur complements with structural zeros, odd panel counts and tails that cut
    through supernodes; solve replay and device-refactorisation replay must still agree with the plain factor."""
    monkeypatch.setenv("MI_OSQP_DENSE_TAIL", str(k))
    P, (l, A, u), _ = PR.gomp_qp(6, 50, np.zeros(6), np.ones(6))
    assert _kkt_check(P, A, l, u)["dense_tail_rows"] == k
    dL, dD, _ = M.debug_host_block_factor(P, A, l, u)
    assert dL <= 1e-9 and dD <= 1e-9
    pr = PR.random_box_qp(1, n=96, mg=64, nnz_per_row=6)
    P, A = PR.qp_matrices(pr, 0)
    assert _kkt_check(P, A, pr["l"][0], pr["u"][0], scaling=0)["dense_tail_rows"] == k
    dL, dD, _ = M.debug_host_block_factor(P, A, pr["l"][0], pr["u"][0])
    assert dL <= 1e-9 and dD <= 1e-9
