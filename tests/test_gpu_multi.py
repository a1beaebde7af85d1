"""GPU (-m gpu): the multi-GPU layer on the one-GPU box - the nccl (= RCCL) path at world_size 1, bench.py's own
launcher, the C-ABI's device-list batch (two shards on one device) - plus per-QP failure isolation and the loop-order
corner at max_iter.  Everything through the C-ABI, checked against the oracle."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import osqp_solver_amd as M
from oracle import oracle as O
from osqp_solver_amd import problems as PR

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ST2EXIT = {1: 0, -3: 1, -4: 2, 2: 3, 3: 4, 4: 5, -2: 6, -7: 9, -10: 10}


def _last_json(text):
    lines = [ln for ln in text.splitlines() if ln.startswith("{")]
    assert lines, text[-2000:]
    return json.loads(lines[-1])


def test_bench_under_torchrun_nccl_world_size_1():
    """The product's multi-GPU path (BatchSolver(device=local_rank) + RCCL gather of solutions) through the exact launch
    line the driver uses, at the one world size this box allows."""
    env = dict(os.environ, MI_OSQP_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29600 + (os.getpid() % 300)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1",
           "--batch", "64", "--no-cpu-baseline", "--no-secondary"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    out = _last_json(res.stdout)
    assert out["n_gpus"] == 1 and out["all_solved"] and out["gather"]["round_trip_ok"] is True
    assert out["config"]["total_qps_per_step"] == 64 and out["roofline"]["frac"] > 0


def test_bench_refuses_to_report_fewer_gpus_than_requested():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 GPUs")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True,
                         text=True, timeout=600, cwd=ROOT)
    assert res.returncode != 0 and "only 1 GPU(s) are visible" in res.stderr and not [ln for ln in res.stdout.splitlines() if ln.startswith("{")]


def test_multi_device_abi_two_shards_match_single_handle_and_oracle():
    B = 7                                                   # ragged: 4 + 3
    pr = PR.random_box_qp(B, n=96, mg=64, nnz_per_row=6)
    one = M.BatchSolver(pr["P"], pr["Px"], pr["q"], pr["A"], pr["Ax"], pr["l"], pr["u"])
    i1 = one.solve(); x1 = one.primal()
    multi = M.MultiBatchSolver(pr["P"], pr["Px"], pr["q"], pr["A"], pr["Ax"], pr["l"], pr["u"], devices=(0, 0))
    assert multi.shards() == [(0, 0, 4), (0, 4, 7)]
    i2 = multi.solve(); x2 = multi.primal()
    assert [(i.status_val, i.iter, i.rho_updates) for i in i1] == [(i.status_val, i.iter, i.rho_updates) for i in i2]
    assert np.array_equal(x1, x2)                           # same kernels, same tiling, same order: bitwise
    assert np.array_equal(one.dual(), multi.dual())
    for b in range(B):
        P, A = PR.qp_matrices(pr, b)
        o = O.OracleQPSolver(P, pr["q"][b], A, pr["l"][b], pr["u"][b])
        st, xo = o.solve()
        assert (i2[b].status_val, i2[b].iter) == (st, o.info().iter) and np.max(np.abs(x2[b] - xo)) <= 1e-6
    # the reference's update -> solve sequence through the sharded handle
    multi.update_bounds(pr["l"] * 0.5, pr["u"] * 0.5); one.update_bounds(pr["l"] * 0.5, pr["u"] * 0.5)
    multi.warm_start_x(x2); one.warm_start_x(x1)
    j2 = multi.solve(); j1 = one.solve()
    assert [(i.status_val, i.iter) for i in j1] == [(i.status_val, i.iter) for i in j2] and np.array_equal(one.primal(), multi.primal())
    # QPSolver::update as one call (new A values + bounds) through the sharded handle
    Ax2 = pr["Ax"] * 1.05
    multi.update_A_bounds(Ax2, pr["l"] * 0.7, pr["u"] * 0.7); one.update_A_bounds(Ax2, pr["l"] * 0.7, pr["u"] * 0.7)
    k2 = multi.solve(); k1 = one.solve()
    assert [(i.status_val, i.iter) for i in k1] == [(i.status_val, i.iter) for i in k2] and np.array_equal(one.primal(), multi.primal())
    # solve_async / wait on the shards' long-lived workers: the same solve, the caller free in between; a getter joins too
    multi.update_bounds(pr["l"] * 0.6, pr["u"] * 0.6); one.update_bounds(pr["l"] * 0.6, pr["u"] * 0.6)
    multi.solve_async()
    m1 = one.solve()
    m2 = multi.wait()
    assert [(i.status_val, i.iter) for i in m1] == [(i.status_val, i.iter) for i in m2] and np.array_equal(one.primal(), multi.primal())
    multi.solve_async(); one.solve()
    assert np.array_equal(one.primal(), multi.primal())     # (get_primal joins the pending solve first)


def test_one_nonconvex_qp_is_isolated_from_the_batch():
    """[REF] src/osqp-wrapper.h:51-54: solve() never throws and returns one exit code per solver.  A batch with one
    indefinite P: that QP reports kNonConvex (NaN solution), the other seven match the oracle exactly."""
    B, bad = 8, 3
    pr = PR.random_box_qp(B, n=96, mg=64, nnz_per_row=6)
    pr["Px"][bad] = -np.abs(pr["Px"][bad]) - 1.0           # same pattern, negative definite-ish: KKT inertia is wrong
    s = M.BatchSolver(pr["P"], pr["Px"], pr["q"], pr["A"], pr["Ax"], pr["l"], pr["u"])
    for rep in range(2):                                   # a second solve reports the same
        info = s.solve(); x = s.primal(); y = s.dual()
        for b in range(B):
            if b == bad:
                assert info[b].status_val == -7 and info[b].exit_code == ST2EXIT[-7] and M.EXIT_NAMES[info[b].exit_code] == "kNonConvex"
                assert np.all(np.isnan(x[b])) and np.all(np.isnan(y[b]))
                continue
            P, A = PR.qp_matrices(pr, b)
            o = O.OracleQPSolver(P, pr["q"][b], A, pr["l"][b], pr["u"][b])
            st, xo = o.solve()
            if rep == 1:
                st, xo = o.solve()
            assert (info[b].status_val, info[b].iter, info[b].rho_updates) == (st, o.info().iter, o.info().rho_updates)
            assert info[b].exit_code == 0 and np.max(np.abs(x[b] - xo)) <= 1e-6
    # bounds update (no refactorisation of the failed QP happens): still isolated
    s.update_bounds(pr["l"] * 0.5, pr["u"] * 0.5)
    info = s.solve()
    assert [i.status_val == -7 for i in info] == [b == bad for b in range(B)]
    # the oracle refuses the same QP at setup
    P, A = PR.qp_matrices(pr, bad)
    with pytest.raises(Exception):
        O.OracleQPSolver(P, pr["q"][bad], A, pr["l"][bad], pr["u"][bad])
    # a single-QP handle keeps reporting the setup error (OsqpSolver::Init fails)
    with pytest.raises(M.MiOsqpError) as e:
        M.BatchSolver(pr["P"], pr["Px"][bad:bad + 1], pr["q"][bad:bad + 1], pr["A"], pr["Ax"][bad:bad + 1], pr["l"][bad:bad + 1], pr["u"][bad:bad + 1])
    assert e.value.code == 4


@pytest.mark.parametrize("check", [0, 30])
def test_max_iter_on_a_rho_iteration_that_is_not_a_check_iteration(check):
    """Upstream's loop order at iter == max_iter on a rho-update iteration without a termination check (check_termination
    = 0, or max_iter not a multiple of it): rho adapts (and the factor is rebuilt) BEFORE the closing check, even when
    that check then reports 'solved'; the next warm-started Solve() continues from that factor."""
    kw = dict(check_termination=check, adaptive_rho_interval=50, max_iter=50, eps_abs=1e-2, eps_rel=1e-2)
    B = 6
    pr = PR.random_box_qp(B, n=96, mg=64, nnz_per_row=6)
    s = M.BatchSolver(pr["P"], pr["Px"], pr["q"], pr["A"], pr["Ax"], pr["l"], pr["u"], **kw)
    i1 = s.solve(); x1 = s.primal().copy()
    i2 = s.solve(); x2 = s.primal().copy()
    seen_update = False
    for b in range(B):
        P, A = PR.qp_matrices(pr, b)
        o = O.OracleQPSolver(P, pr["q"][b], A, pr["l"][b], pr["u"][b], **kw)
        for info, x in ((i1, x1), (i2, x2)):
            st, xo = o.solve(); io = o.info()
            assert (info[b].status_val, info[b].iter, info[b].rho_updates) == (st, io.iter, io.rho_updates), (b, check)
            assert abs(info[b].rho - io.rho) <= 1e-9 * io.rho and abs(info[b].rho_estimate - io.rho_estimate) <= 1e-6 * io.rho_estimate
            tol = 1e-6 if st in (1, 2) else 1e-5
            assert np.max(np.abs(x[b] - xo)) <= tol
            seen_update = seen_update or io.rho_updates > 0
    assert seen_update                                       # the corner is really exercised


# ---------------------------------------------------------------- co-residency of the grid-spinning kernels
def _grid_parity(pr, s):
    from oracle import oracle as O
    info = s.solve()
    P, A = PR.qp_matrices(pr, 0)
    o = O.OracleQPSolver(P, pr["q"][0], A, pr["l"][0], pr["u"][0])
    st, xo = o.solve()
    assert info[0].status_val == st and info[0].iter == o.info().iter, (info[0].status_val, st, info[0].iter, o.info().iter)
    assert np.max(np.abs(s.primal()[0] - xo)) <= 1e-6


def test_oversized_group_grid_is_clamped_to_what_stays_resident(monkeypatch):
    """The dataflow sweeps of a large single QP wait for each other's workgroups: the grid is clamped at setup to what the
    device keeps resident (occupancy of the spinning kernels x CUs).  On 256 CUs every legal grid fits, so the test lets
    the clamp assume a 4-CU device: 256 requested workgroups of 512 threads must come down, and the solve must be right."""
    monkeypatch.setenv("MI_OSQP_GROUPS", "256")
    monkeypatch.setenv("MI_OSQP_GROUP_THREADS", "512")
    monkeypatch.setenv("MI_OSQP_ASSUME_CUS", "4")
    pr = PR.grid_qp(90)
    s = M.BatchSolver(pr["P"], pr["Px"], pr["q"], pr["A"], pr["Ax"], pr["l"], pr["u"])
    st = s.stats()
    assert 2 <= st["solve_groups"] <= 4 * 4 and st["solve_group_threads"] == 512, st      # (at most 4 workgroups of 512 threads per CU)
    _grid_parity(pr, s)
    monkeypatch.delenv("MI_OSQP_ASSUME_CUS")
    s2 = M.BatchSolver(pr["P"], pr["Px"], pr["q"], pr["A"], pr["Ax"], pr["l"], pr["u"])
    assert s2.stats()["solve_groups"] == 256


def test_two_dataflow_handles_on_one_device_take_turns():
    """Two large single QPs on ONE device - as two shards of the device-list API and from two host threads: their
    grid-spinning launches are serialised per device, so neither starves the other; both match the oracle."""
    import threading
    pr = PR.grid_qp(90)
    pr2 = {k: (np.concatenate([v, v]) if isinstance(v, np.ndarray) and v.ndim == 2 else v) for k, v in pr.items()}
    pr2["l"] = pr2["l"].copy(); pr2["l"][1] *= 0.9
    ms = M.MultiBatchSolver(pr2["P"], pr2["Px"], pr2["q"], pr2["A"], pr2["Ax"], pr2["l"], pr2["u"], devices=(0, 0))
    assert [(b, e) for _, b, e in ms.shards()] == [(0, 1), (1, 2)]
    info = ms.solve()
    x = ms.primal()
    from oracle import oracle as O
    for b in range(2):
        P, A = PR.qp_matrices(pr2, b)
        o = O.OracleQPSolver(P, pr2["q"][b], A, pr2["l"][b], pr2["u"][b])
        st, xo = o.solve()
        assert info[b].status_val == st and info[b].iter == o.info().iter and np.max(np.abs(x[b] - xo)) <= 1e-6
    solvers = [M.BatchSolver(pr["P"], pr["Px"], pr["q"], pr["A"], pr["Ax"], pr["l"], pr["u"]) for _ in range(2)]
    assert all(s.stats()["solve_groups"] > 1 for s in solvers)
    errs = []

    def work(s):
        try:
            for _ in range(3):
                s.reset()
                _grid_parity(pr, s)
        except Exception as e:      # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=work, args=(s,)) for s in solvers]
    [t.start() for t in th]; [t.join() for t in th]
    assert not errs, errs
    assert np.array_equal(solvers[0].primal(), solvers[1].primal())
