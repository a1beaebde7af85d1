"""include/osqp++.h: the osqp-cpp-shaped shim that lets the reference's headers compile UNCHANGED on the MI355X C-ABI.

Eigen is not in this container, so the compile uses tests/cpp/eigen_standin (a ~100-line syntax stand-in: it pins
nothing about Eigen).  The reference's own src/osqp-wrapper.h is included from /root/reference where that exists (the
build container) and never copied; on the GPU box the same call sequence runs through osqp::OsqpSolver directly and is
compared with the ctypes binding of the same library."""
import json
import os
import subprocess

import numpy as np
import pytest
import scipy.sparse as sp

import osqp_solver_amd as M

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_WRAPPER = "/root/reference/src/osqp-wrapper.h"


def _build(tmp_path, with_ref):
    M.lib()
    exe = str(tmp_path / ("shim_ref" if with_ref else "shim_direct"))
    cmd = ["g++", "-std=c++17", "-O1", "-DNDEBUG", "-I", os.path.join(ROOT, "tests", "cpp", "eigen_standin"),
           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "osqp_shim_test.cpp"),
           "-L", os.path.join(ROOT, "osqp-solver_amd"), "-lmi_osqp", "-Wl,-rpath," + os.path.join(ROOT, "osqp-solver_amd"), "-o", exe]
    if with_ref:
        cmd.insert(1, f'-DMI_REF_WRAPPER="{REF_WRAPPER}"')
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    return exe


def _run(exe):
    res = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    return json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1]), res.stdout


def test_shim_compiles_and_reports_errors_like_osqp_cpp_without_a_gpu(tmp_path, gpu_available):
    if gpu_available:
        pytest.skip("a GPU is present")
    out, log = _run(_build(tmp_path, False))
    assert out["code1"] == "kUnknown" and out["threw"] is True          # Init failed (no device) -> Solve() == kUnknown
    assert "FAILED_PRECONDITION: OsqpSolver is not initialized." in log


def test_reference_wrapper_compiles_unchanged_against_the_shim(tmp_path, gpu_available):
    """[REF] src/osqp-wrapper.h included as it is (build container only)."""
    if not os.path.exists(REF_WRAPPER):
        pytest.skip("the reference is not on this machine")
    out, log = _run(_build(tmp_path, True))
    assert "3, 3, 2, 3, 2, 2" in log                                     # the wrapper's own constructor print ([REF] :19)
    if not gpu_available:
        assert out["code1"] == "kUnknown"


@pytest.mark.gpu
def test_shim_call_sequence_matches_the_ctypes_binding(tmp_path):
    out, log = _run(_build(tmp_path, False))
    P = sp.csc_matrix(np.array([[4.0, 1.0], [1.0, 2.0]])); A = sp.csc_matrix(np.array([[1.0, 1.0], [1.0, 0.0], [0.0, 1.0]]))
    l = np.array([1.0, 0.0, 0.0]); u = np.array([1.0, 0.7, 0.7])
    s = M.QPSolver((l, A, u), P)
    s.setWarmStart(np.array([0.3, 0.7]))
    c1, x1 = s.solve(); it1 = s.info().iter
    s.update((l, A, np.array([1.0, 0.6, 0.9])))
    c2, x2 = s.solve(); it2 = s.info().iter
    assert out["threw"] is False and "Init: OK" in log and "STATUS: OK" in log
    assert (out["code1"], out["code2"]) == (M.EXIT_NAMES[c1], M.EXIT_NAMES[c2]) == ("kOptimal", "kOptimal")
    assert (out["it1"], out["it2"]) == (it1, it2)
    assert np.array_equal(np.array(out["x1"]), x1) and np.array_equal(np.array(out["x2"]), x2)      # same library, same kernels: bitwise
