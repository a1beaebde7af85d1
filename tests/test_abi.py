"""CPU: the C-ABI library loads, exports every symbol include/mi_osqp.h declares,
and refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import scipy.sparse as sp

import osqp_solver_amd as M

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "mi_osqp.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mi_(?:osqp|gomp)_[a-z_A-Z0-9]+)\s*\(", src)))


def test_every_declared_symbol_is_exported():
    L = M.lib()
    names = _declared()
    assert len(names) >= 30 and "mi_gomp_relinearise_some" in names and "mi_osqp_prefetch_analysis" in names
    for nme in names:
        assert hasattr(L, nme), f"{nme} declared in include/mi_osqp.h but not exported"


def test_defaults_are_osqp_defaults():
    s = M.default_settings()
    assert (s.rho, s.sigma, s.scaling, s.alpha) == (0.1, 1e-6, 10, 1.6)
    assert (s.max_iter, s.eps_abs, s.eps_rel, s.check_termination) == (4000, 1e-3, 1e-3, 25)
    assert (s.eps_prim_inf, s.eps_dual_inf, s.adaptive_rho, s.adaptive_rho_tolerance) == (1e-4, 1e-4, 1, 5.0)
    assert s.warm_start == 1 and s.scaled_termination == 0


def test_exit_code_names_follow_osqp_cpp():
    L = M.lib()
    assert [L.mi_osqp_exit_code_name(i).decode() for i in range(11)] == M.EXIT_NAMES
    assert L.mi_osqp_exit_code_name(99).decode() == "kUnknown"


def test_invalid_input_codes_need_no_gpu():
    P = sp.eye(2).tocsc(); A = sp.eye(2).tocsc()
    with pytest.raises(M.MiOsqpError) as e:
        M.BatchSolver(P, P.data, None, A, A.data, [1.0, 0.0], [0.0, 1.0])       # l > u
    assert e.value.code == 1
    with pytest.raises(M.MiOsqpError) as e:
        M.BatchSolver(P, P.data, None, A, A.data, [0.0, 0.0], [1.0, 1.0], alpha=3.0)
    assert e.value.code == 2


def test_no_cpu_fallback_without_gpu(gpu_available):
    if gpu_available:
        pytest.skip("a GPU is present")
    P = sp.eye(2).tocsc(); A = sp.eye(2).tocsc()
    with pytest.raises(M.MiOsqpError) as e:
        M.BatchSolver(P, P.data, None, A, A.data, [0.0, 0.0], [1.0, 1.0])
    assert e.value.code == 5     # MI_OSQP_ERR_DEVICE: the product never solves on the CPU


def test_product_does_not_reference_oracle():
    pkg = os.path.join(ROOT, "osqp-solver_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "osqp_oracle" not in txt, f
    for f in ("include/mi_osqp.h",):
        assert "osqp_oracle" not in open(os.path.join(ROOT, f)).read()


def test_prefetch_analysis_fills_the_cache_without_a_gpu():
    """mi_osqp_prefetch_analysis is host work only: it runs here, a second call with the same pattern is a cache hit (much
    faster), concurrent calls for one pattern all succeed (one computes, the others wait), bad data is refused."""
    import threading, time
    from osqp_solver_amd import problems as PR
    pr = PR.gomp_batch(1, 6, 120)
    P, A = PR.qp_matrices(pr, 0)
    t = time.perf_counter(); assert M.prefetch_analysis(P, A) == 0; t_first = time.perf_counter() - t
    t = time.perf_counter(); assert M.prefetch_analysis(P, A) == 0; t_again = time.perf_counter() - t
    assert t_again < 0.5 * t_first + 1e-3, (t_first, t_again)
    pr2 = PR.gomp_batch(1, 6, 130)
    P2, A2 = PR.qp_matrices(pr2, 0)
    rcs = []
    th = [threading.Thread(target=lambda: rcs.append(M.prefetch_analysis(P2, A2))) for _ in range(4)]
    [x.start() for x in th]; [x.join() for x in th]
    assert rcs == [0, 0, 0, 0]
    bad = A.copy(); bad.indices = bad.indices.copy(); bad.indices[0] = A.shape[0] + 5
    assert M.prefetch_analysis(P, bad, check=False) != 0
