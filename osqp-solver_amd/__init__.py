"""osqp-solver_amd -- MI355X-native OSQP ADMM core behind the reference's QPSolver API.

Python here is only the test/bench harness language (the reference is C++; its
drop-in is include/mi_osqp.h + include/mi_osqp/qp_solver.hpp).  This module is
a ctypes binding of the C-ABI; it never computes anything itself and never
touches oracle/.  If libmi_osqp.so cannot be built/loaded, import fails loudly.
"""
import ctypes as C
import os

import numpy as np

# torch bundles its own HIP runtime (same SONAME as /opt/rocm's).  Whichever is
# loaded first wins for the whole process, and torch cannot initialise on top of
# the system one -- so when torch is installed it must be imported before
# libmi_osqp.so is dlopen'ed.  torch is plumbing here (device tensors, streams,
# torch.distributed); the library itself does not need it.
try:
    import torch  # noqa: F401
except ImportError:  # pragma: no cover
    torch = None

from . import build as _build

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

EXIT_NAMES = ["kOptimal", "kPrimalInfeasible", "kDualInfeasible", "kOptimalInaccurate",
              "kPrimalInfeasibleInaccurate", "kDualInfeasibleInaccurate", "kMaxIterations",
              "kInterrupted", "kTimeLimitReached", "kNonConvex", "kUnknown"]
K_OPTIMAL = 0


class Settings(C.Structure):
    _fields_ = [("rho", C.c_double), ("sigma", C.c_double), ("scaling", C.c_int64),
                ("adaptive_rho", C.c_int64), ("adaptive_rho_interval", C.c_int64),
                ("adaptive_rho_tolerance", C.c_double), ("max_iter", C.c_int64),
                ("eps_abs", C.c_double), ("eps_rel", C.c_double), ("eps_prim_inf", C.c_double),
                ("eps_dual_inf", C.c_double), ("alpha", C.c_double), ("scaled_termination", C.c_int64),
                ("check_termination", C.c_int64), ("warm_start", C.c_int64), ("verbose", C.c_int64)]


class Info(C.Structure):
    _fields_ = [("iter", C.c_int64), ("status_val", C.c_int64), ("exit_code", C.c_int64),
                ("obj_val", C.c_double), ("pri_res", C.c_double), ("dua_res", C.c_double),
                ("rho_updates", C.c_int64), ("rho_estimate", C.c_double), ("rho", C.c_double)]


class Stats(C.Structure):
    _fields_ = [(k, C.c_int64) for k in
                ("n", "m", "N", "batch", "tile", "n_tiles", "nnz_P_triu", "nnz_A", "nnz_KKT", "nnz_L",
                 "n_supernodes", "n_blocks", "fwd_levels", "bwd_levels", "fwd_slots", "bwd_slots",
                 "chk_slots", "lds_bytes", "threads_per_block", "dense_tail_rows", "dense_tail_slots")] + \
               [(k, C.c_double) for k in ("setup_seconds_host", "setup_seconds_factor", "setup_seconds_upload")] + \
               [("nnz_L_before_tail", C.c_int64), ("solve_groups", C.c_int64), ("solve_group_threads", C.c_int64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class MiOsqpError(RuntimeError):
    def __init__(self, code, where):
        self.code = int(code)
        L = lib()
        msg = L.mi_osqp_error_name(int(code)).decode()
        extra = L.mi_osqp_last_error().decode()
        super().__init__(f"{where}: error {int(code)} ({msg}) {extra}")


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libmi_osqp.so")
        if os.environ.get("MI_OSQP_LIBRARY"):      # (tests: the diagnostic twin of the library, build.build_debug)
            path = os.environ["MI_OSQP_LIBRARY"]
        else:
            try:
                path = _build.build()
            except Exception as e:  # hipcc missing on this machine: use the shipped .so or fail
                if not os.path.exists(path):
                    raise ImportError(f"libmi_osqp.so missing and could not be built: {e}") from e
        L = C.CDLL(path)
        ip, dp, vp = C.POINTER(C.c_int64), C.POINTER(C.c_double), C.c_void_p
        L.mi_osqp_default_settings.argtypes = [C.POINTER(Settings)]
        for f in ("mi_osqp_exit_code_name", "mi_osqp_error_name"):
            getattr(L, f).restype = C.c_char_p; getattr(L, f).argtypes = [C.c_int64]
        L.mi_osqp_version.restype = C.c_char_p
        L.mi_osqp_last_error.restype = C.c_char_p
        L.mi_osqp_batch_setup.argtypes = [C.POINTER(vp), C.c_int64, C.c_int64, C.c_int64, ip, ip, dp, dp, ip, ip, dp,
                                          dp, dp, C.POINTER(Settings), C.c_int64]
        L.mi_osqp_batch_update_A.argtypes = [vp, ip, ip, dp]
        L.mi_osqp_batch_update_A_bounds.argtypes = [vp, ip, ip, dp, dp, dp]
        L.mi_osqp_batch_update_bounds.argtypes = [vp, dp, dp]
        L.mi_osqp_batch_warm_start_x.argtypes = [vp, dp]
        L.mi_osqp_batch_solve.argtypes = [vp]
        L.mi_osqp_batch_get_primal.argtypes = [vp, dp]
        L.mi_osqp_batch_get_dual.argtypes = [vp, dp]
        L.mi_osqp_batch_get_info.argtypes = [vp, C.POINTER(Info)]
        L.mi_osqp_batch_get_stats.argtypes = [vp, C.POINTER(Stats)]
        L.mi_osqp_batch_get_ordering.argtypes = [vp, ip]
        L.mi_osqp_batch_free.argtypes = [vp]; L.mi_osqp_batch_free.restype = None
        L.mi_osqp_batch_update_bounds_device.argtypes = [vp, vp, vp, vp]
        L.mi_osqp_batch_update_A_bounds_device.argtypes = [vp, vp, vp, vp, vp]
        L.mi_osqp_batch_solve_device.argtypes = [vp, vp, vp, vp, vp]
        L.mi_osqp_batch_reset.argtypes = [vp]
        L.mi_osqp_batch_refactor_device.argtypes = [vp]
        L.mi_osqp_batch_last_solve_stats.argtypes = [vp, ip, ip, dp, dp, ip, dp]
        L.mi_osqp_batch_spmv.argtypes = [vp, vp, vp, vp, vp, vp, vp]
        L.mi_osqp_batch_kkt_solve.argtypes = [vp, vp, vp, vp]
        L.mi_osqp_batch_kernel_time.argtypes = [vp, dp, ip]
        L.mi_osqp_batch_refactor_time.argtypes = [vp, dp, dp, ip, ip]
        L.mi_osqp_batch_refactor_peak.argtypes = [vp, ip, dp, dp]
        L.mi_osqp_batch_reinit_some.argtypes = [vp, C.c_int64, ip, dp, dp, dp]
        L.mi_osqp_batch_update_A_bounds_some.argtypes = [vp, C.c_int64, ip, dp, dp, dp]
        L.mi_osqp_batch_warm_start_x_some.argtypes = [vp, C.c_int64, ip, dp]
        L.mi_osqp_batch_solve_begin_some.argtypes = [vp, C.c_int64, ip]
        L.mi_osqp_batch_advance.argtypes = [vp, C.c_int64]
        L.mi_osqp_batch_poll.argtypes = [vp, C.c_int64, ip, ip, C.c_int64]
        L.mi_osqp_batch_get_primal_some.argtypes = [vp, C.c_int64, ip, dp]
        L.mi_osqp_batch_get_dual_some.argtypes = [vp, C.c_int64, ip, dp]
        L.mi_osqp_batch_get_info_some.argtypes = [vp, C.c_int64, ip, C.POINTER(Info)]
        L.mi_osqp_batch_running.argtypes = [vp]; L.mi_osqp_batch_running.restype = C.c_int64
        L.mi_osqp_setup.argtypes = [C.POINTER(vp), C.c_int64, C.c_int64, ip, ip, dp, dp, ip, ip, dp, dp, dp, C.POINTER(Settings)]
        L.mi_osqp_update_A.argtypes = [vp, ip, ip, dp]
        L.mi_osqp_update_bounds.argtypes = [vp, dp, dp]
        L.mi_osqp_warm_start_x.argtypes = [vp, dp]
        L.mi_osqp_solve.argtypes = [vp, C.POINTER(Info)]
        L.mi_osqp_get_primal.argtypes = [vp, dp]
        L.mi_osqp_get_dual.argtypes = [vp, dp]
        L.mi_osqp_get_stats.argtypes = [vp, C.POINTER(Stats)]
        L.mi_osqp_free.argtypes = [vp]; L.mi_osqp_free.restype = None
        L.mi_osqp_multi_batch_setup.argtypes = [C.POINTER(vp), C.c_int64, ip, C.c_int64, C.c_int64, C.c_int64, ip, ip, dp, dp,
                                                ip, ip, dp, dp, dp, C.POINTER(Settings)]
        L.mi_osqp_multi_batch_update_A.argtypes = [vp, ip, ip, dp]
        L.mi_osqp_multi_batch_update_bounds.argtypes = [vp, dp, dp]
        L.mi_osqp_multi_batch_update_A_bounds.argtypes = [vp, ip, ip, dp, dp, dp]
        L.mi_osqp_multi_batch_warm_start_x.argtypes = [vp, dp]
        L.mi_osqp_multi_batch_solve.argtypes = [vp]
        L.mi_osqp_multi_batch_solve_async.argtypes = [vp]
        L.mi_osqp_multi_batch_wait.argtypes = [vp]
        L.mi_osqp_multi_batch_get_primal.argtypes = [vp, dp]
        L.mi_osqp_multi_batch_get_dual.argtypes = [vp, dp]
        L.mi_osqp_multi_batch_get_info.argtypes = [vp, C.POINTER(Info)]
        L.mi_osqp_multi_batch_shards.argtypes = [vp]; L.mi_osqp_multi_batch_shards.restype = C.c_int64
        L.mi_osqp_multi_batch_shard.argtypes = [vp, C.c_int64, ip, ip, ip, C.POINTER(vp)]
        L.mi_osqp_multi_batch_free.argtypes = [vp]; L.mi_osqp_multi_batch_free.restype = None
        L.mi_osqp_debug_host_kkt_solve.argtypes = [C.c_int64, C.c_int64, ip, ip, dp, ip, ip, dp, dp, dp,
                                                   C.POINTER(Settings), C.c_int64, dp, dp, dp, C.POINTER(Stats)]
        L.mi_osqp_prefetch_analysis.argtypes = [C.c_int64, C.c_int64, C.c_int64, ip, ip, ip, ip, C.c_int64]
        L.mi_osqp_debug_host_block_factor.argtypes = [C.c_int64, C.c_int64, ip, ip, dp, ip, ip, dp, dp, dp,
                                                      C.POINTER(Settings), dp, dp, ip]
        _LIB = L
    return _LIB


def host_cores():
    """CPU cores this process may actually use: min(affinity, cgroup quota)."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != "max":
                n = min(n, max(1, int(round(int(quota) / int(period)))))
        except Exception:
            pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            n = min(n, max(1, int(round(q / p))))
    except Exception:
        pass
    return n


def default_settings(**kw):
    s = Settings()
    lib().mi_osqp_default_settings(C.byref(s))
    for k, v in kw.items():
        if not hasattr(s, k):
            raise KeyError(k)
        setattr(s, k, v)
    return s


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _chk(rc, where):
    if rc != 0:
        raise MiOsqpError(rc, where)


def _csc(M):
    import scipy.sparse as sp
    M = sp.csc_matrix(M)
    M.sort_indices()
    return M


class BatchSolver:
    """B QPs with one shared sparsity pattern (P_pattern / A_pattern are scipy
    sparse matrices carrying the pattern; Px[B,nnzP], Ax[B,nnzA] the per-QP values
    in CSC order of those patterns)."""

    def __init__(self, P_pattern, Px, q, A_pattern, Ax, l, u, device=-1, **settings):
        L = lib()
        P, A = _csc(P_pattern), _csc(A_pattern)
        self.n, self.m = A.shape[1], A.shape[0]
        Px, Ax, l, u = _f64(Px), _f64(Ax), _f64(l), _f64(u)
        if Ax.ndim == 1:
            Px, Ax, l, u = Px[None], Ax[None], l[None], u[None]
            q = None if q is None else _f64(q)[None]
        self.B = Ax.shape[0]
        q = None if q is None else _f64(q)
        self._Pp, self._Pi = _i64(P.indptr), _i64(P.indices)
        self._Ap, self._Ai = _i64(A.indptr), _i64(A.indices)
        self.settings = default_settings(**settings)
        self._h = C.c_void_p()
        rc = L.mi_osqp_batch_setup(C.byref(self._h), self.B, self.n, self.m, _ip(self._Pp), _ip(self._Pi), _dp(Px),
                                   _dp(q), _ip(self._Ap), _ip(self._Ai), _dp(Ax), _dp(l), _dp(u),
                                   C.byref(self.settings), device)
        _chk(rc, "mi_osqp_batch_setup")

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                lib().mi_osqp_batch_free(self._h)
                self._h = C.c_void_p()
        except Exception:       # interpreter shutdown: module globals may already be gone
            pass

    close = __del__

    def solve(self):
        _chk(lib().mi_osqp_batch_solve(self._h), "mi_osqp_batch_solve")
        return self.info()

    def primal(self):
        x = np.empty((self.B, self.n))
        _chk(lib().mi_osqp_batch_get_primal(self._h, _dp(x)), "get_primal")
        return x

    def dual(self):
        y = np.empty((self.B, self.m))
        _chk(lib().mi_osqp_batch_get_dual(self._h, _dp(y)), "get_dual")
        return y

    def info(self):
        arr = (Info * self.B)()
        _chk(lib().mi_osqp_batch_get_info(self._h, arr), "get_info")
        return list(arr)

    def stats(self):
        s = Stats()
        _chk(lib().mi_osqp_batch_get_stats(self._h, C.byref(s)), "get_stats")
        return s.as_dict()

    def ordering(self):
        """Elimination order of the KKT matrix chosen by the analysis (natural index eliminated k-th)."""
        perm = np.empty(self.n + self.m, dtype=np.int64)
        _chk(lib().mi_osqp_batch_get_ordering(self._h, _ip(perm)), "get_ordering")
        return perm

    def update_A(self, Ax, A_pattern=None):
        Ax = _f64(Ax).reshape(self.B, -1)
        Ap, Ai = self._Ap, self._Ai
        if A_pattern is not None:
            A = _csc(A_pattern)
            Ap, Ai = _i64(A.indptr), _i64(A.indices)
            if len(Ai) != len(self._Ai):
                raise MiOsqpError(3, "update_A")
        _chk(lib().mi_osqp_batch_update_A(self._h, _ip(Ap), _ip(Ai), _dp(Ax)), "update_A")

    def update_A_bounds(self, Ax, l, u):
        """QPSolver::update as one call: new A values and new bounds, one refactorisation."""
        Ax = _f64(Ax).reshape(self.B, -1)
        l, u = _f64(l).reshape(self.B, -1), _f64(u).reshape(self.B, -1)
        _chk(lib().mi_osqp_batch_update_A_bounds(self._h, _ip(self._Ap), _ip(self._Ai), _dp(Ax), _dp(l), _dp(u)), "update_A_bounds")

    def update_bounds(self, l, u):
        l, u = _f64(l).reshape(self.B, -1), _f64(u).reshape(self.B, -1)
        _chk(lib().mi_osqp_batch_update_bounds(self._h, _dp(l), _dp(u)), "update_bounds")

    def warm_start_x(self, x):
        x = _f64(x).reshape(self.B, -1)
        _chk(lib().mi_osqp_batch_warm_start_x(self._h, _dp(x)), "warm_start_x")

    def reset(self):
        _chk(lib().mi_osqp_batch_reset(self._h), "reset")

    def refactor_device(self):
        _chk(lib().mi_osqp_batch_refactor_device(self._h), "refactor_device")

    def last_solve_stats(self):
        it, ln, rc = C.c_int64(), C.c_int64(), C.c_int64()
        ds, rs, cs = C.c_double(), C.c_double(), C.c_double()
        _chk(lib().mi_osqp_batch_last_solve_stats(self._h, C.byref(it), C.byref(ln), C.byref(ds), C.byref(rs), C.byref(rc),
                                                  C.byref(cs)), "stats")
        return dict(total_iters=it.value, launches=ln.value, device_s=ds.value, refactor_s=rs.value, refactors=rc.value,
                    compact_s=cs.value)

    def kernel_time(self):
        ms, cnt = C.c_double(), C.c_int64()
        _chk(lib().mi_osqp_batch_kernel_time(self._h, C.byref(ms), C.byref(cnt)), "kernel_time")
        return ms.value, cnt.value

    def refactor_time(self):
        """(factor_kernel ms, dense_inverse_kernel ms, launches, QPs refactored) since the last call."""
        f, d, ln, nq = C.c_double(), C.c_double(), C.c_int64(), C.c_int64()
        _chk(lib().mi_osqp_batch_refactor_time(self._h, C.byref(f), C.byref(d), C.byref(ln), C.byref(nq)), "refactor_time")
        return f.value, d.value, ln.value, nq.value

    def refactor_peak(self):
        """(QPs, factor_kernel ms, dense-tail kernels ms) of the largest refactorisation since the last refactor_time()."""
        nq, f, d = C.c_int64(), C.c_double(), C.c_double()
        _chk(lib().mi_osqp_batch_refactor_peak(self._h, C.byref(nq), C.byref(f), C.byref(d)), "refactor_peak")
        return nq.value, f.value, d.value

    # ---- continuous batching: per-QP entry points + non-blocking advance / poll (mi_osqp.h "continuous batching")
    def reinit_some(self, ids, Ax, l, u):
        ids = _i64(ids); k = len(ids)
        Ax, l, u = _f64(Ax).reshape(k, -1), _f64(l).reshape(k, -1), _f64(u).reshape(k, -1)
        _chk(lib().mi_osqp_batch_reinit_some(self._h, k, _ip(ids), _dp(Ax), _dp(l), _dp(u)), "reinit_some")

    def update_A_bounds_some(self, ids, Ax, l, u):
        ids = _i64(ids); k = len(ids)
        Ax, l, u = _f64(Ax).reshape(k, -1), _f64(l).reshape(k, -1), _f64(u).reshape(k, -1)
        _chk(lib().mi_osqp_batch_update_A_bounds_some(self._h, k, _ip(ids), _dp(Ax), _dp(l), _dp(u)), "update_A_bounds_some")

    def warm_start_x_some(self, ids, x):
        ids = _i64(ids); x = _f64(x).reshape(len(ids), -1)
        _chk(lib().mi_osqp_batch_warm_start_x_some(self._h, len(ids), _ip(ids), _dp(x)), "warm_start_x_some")

    def solve_begin_some(self, ids):
        ids = _i64(ids)
        _chk(lib().mi_osqp_batch_solve_begin_some(self._h, len(ids), _ip(ids)), "solve_begin_some")

    def advance(self, n_segments=1):
        _chk(lib().mi_osqp_batch_advance(self._h, n_segments), "advance")

    def poll(self, wait=True):
        """QP ids that finished in the oldest advance not polled yet (None: wait=False and it has not run yet)."""
        out = np.empty(self.B, dtype=np.int64)
        nf = C.c_int64()
        _chk(lib().mi_osqp_batch_poll(self._h, 1 if wait else 0, C.byref(nf), _ip(out), self.B), "poll")
        return None if nf.value < 0 else out[:nf.value].copy()

    def running(self):
        return int(lib().mi_osqp_batch_running(self._h))

    def primal_some(self, ids):
        ids = _i64(ids); x = np.empty((len(ids), self.n))
        _chk(lib().mi_osqp_batch_get_primal_some(self._h, len(ids), _ip(ids), _dp(x)), "get_primal_some")
        return x

    def dual_some(self, ids):
        ids = _i64(ids); y = np.empty((len(ids), self.m))
        _chk(lib().mi_osqp_batch_get_dual_some(self._h, len(ids), _ip(ids), _dp(y)), "get_dual_some")
        return y

    def info_some(self, ids):
        ids = _i64(ids); arr = (Info * len(ids))()
        _chk(lib().mi_osqp_batch_get_info_some(self._h, len(ids), _ip(ids), arr), "get_info_some")
        return list(arr)

    # ---- device-resident variants (torch tensors on the solver's GPU)
    def solve_device(self, x_out=None, status=None, iters=None, stream=None):
        p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        _chk(lib().mi_osqp_batch_solve_device(self._h, p(x_out), p(status), p(iters),
                                              None if stream is None else C.c_void_p(stream)), "solve_device")

    def update_A_bounds_device(self, Ax, l, u, stream=None):
        """QPSolver::update from torch CUDA tensors ([B, nnzA], [B, m], [B, m], float64, contiguous)."""
        _chk(lib().mi_osqp_batch_update_A_bounds_device(self._h, C.c_void_p(Ax.data_ptr()), C.c_void_p(l.data_ptr()), C.c_void_p(u.data_ptr()),
                                                        None if stream is None else C.c_void_p(stream)), "update_A_bounds_device")

    def update_bounds_device(self, l, u, stream=None):
        _chk(lib().mi_osqp_batch_update_bounds_device(self._h, C.c_void_p(l.data_ptr()), C.c_void_p(u.data_ptr()),
                                                      None if stream is None else C.c_void_p(stream)), "update_bounds_device")

    def spmv_device(self, x, y, Px, Aty, Ax, stream=None):
        p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        _chk(lib().mi_osqp_batch_spmv(self._h, p(x), p(y), p(Px), p(Aty), p(Ax),
                                      None if stream is None else C.c_void_p(stream)), "spmv")

    def debug_trace_kkt_solve(self, rhs, sol):
        """Diagnostics (tile 2 only): returns (stamps[2 tiles], fwd phase table, bwd phase table, dims)."""
        import numpy as np
        L = lib()
        L.mi_osqp_debug_trace_kkt_solve.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                                    C.POINTER(C.c_int64)]
        dims = (C.c_int64 * 4)()
        _chk(L.mi_osqp_debug_trace_kkt_solve(self._h, 0, None, None, None, 0, dims), "trace dims")
        fp, bp, nw, words = (int(v) for v in dims)
        tr = np.zeros(2 * words, dtype=np.uint32)
        _chk(L.mi_osqp_debug_trace_kkt_solve(self._h, 0, C.c_void_p(rhs.data_ptr()), C.c_void_p(sol.data_ptr()),
                                             tr.ctypes.data_as(C.c_void_p), tr.size, dims), "trace")
        tabs = []
        for which, rows in ((1, fp), (2, bp)):
            t = np.zeros(rows * (4 * nw + 1), dtype=np.uint32)
            _chk(L.mi_osqp_debug_trace_kkt_solve(self._h, which, None, None, t.ctypes.data_as(C.c_void_p), t.size, dims), "trace table")
            tabs.append(t.reshape(rows, 4 * nw + 1))
        return tr.reshape(2, words), tabs[0], tabs[1], (fp, bp, nw, words)

    def kkt_solve_device(self, rhs, sol, stream=None):
        _chk(lib().mi_osqp_batch_kkt_solve(self._h, C.c_void_p(rhs.data_ptr()), C.c_void_p(sol.data_ptr()),
                                           None if stream is None else C.c_void_p(stream)), "kkt_solve")


class MultiBatchSolver:
    """The batch sharded over several HIP devices inside ONE process (mi_osqp_multi_batch_*: block partition, one host
    thread per shard, no data-path collective).  `devices` may list a device more than once."""

    def __init__(self, P_pattern, Px, q, A_pattern, Ax, l, u, devices=(0,), **settings):
        L = lib()
        P, A = _csc(P_pattern), _csc(A_pattern)
        self.n, self.m = A.shape[1], A.shape[0]
        Px, Ax, l, u = _f64(Px), _f64(Ax), _f64(l), _f64(u)
        self.B = Ax.shape[0]
        q = None if q is None else _f64(q)
        self._Pp, self._Pi = _i64(P.indptr), _i64(P.indices)
        self._Ap, self._Ai = _i64(A.indptr), _i64(A.indices)
        self.settings = default_settings(**settings)
        devs = _i64(list(devices))
        self._h = C.c_void_p()
        _chk(L.mi_osqp_multi_batch_setup(C.byref(self._h), len(devs), _ip(devs), self.B, self.n, self.m, _ip(self._Pp),
                                         _ip(self._Pi), _dp(Px), _dp(q), _ip(self._Ap), _ip(self._Ai), _dp(Ax), _dp(l), _dp(u),
                                         C.byref(self.settings)), "mi_osqp_multi_batch_setup")

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                lib().mi_osqp_multi_batch_free(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    close = __del__

    def shards(self):
        out = []
        for k in range(lib().mi_osqp_multi_batch_shards(self._h)):
            d, b, e = C.c_int64(), C.c_int64(), C.c_int64()
            _chk(lib().mi_osqp_multi_batch_shard(self._h, k, C.byref(d), C.byref(b), C.byref(e), None), "shard")
            out.append((d.value, b.value, e.value))
        return out

    def solve(self):
        _chk(lib().mi_osqp_multi_batch_solve(self._h), "mi_osqp_multi_batch_solve")
        return self.info()

    def solve_async(self):
        _chk(lib().mi_osqp_multi_batch_solve_async(self._h), "mi_osqp_multi_batch_solve_async")

    def wait(self):
        _chk(lib().mi_osqp_multi_batch_wait(self._h), "mi_osqp_multi_batch_wait")
        return self.info()

    def info(self):
        arr = (Info * self.B)()
        _chk(lib().mi_osqp_multi_batch_get_info(self._h, arr), "multi get_info")
        return list(arr)

    def primal(self):
        x = np.empty((self.B, self.n))
        _chk(lib().mi_osqp_multi_batch_get_primal(self._h, _dp(x)), "multi get_primal")
        return x

    def dual(self):
        y = np.empty((self.B, self.m))
        _chk(lib().mi_osqp_multi_batch_get_dual(self._h, _dp(y)), "multi get_dual")
        return y

    def update_A(self, Ax):
        Ax = _f64(Ax).reshape(self.B, -1)
        _chk(lib().mi_osqp_multi_batch_update_A(self._h, _ip(self._Ap), _ip(self._Ai), _dp(Ax)), "multi update_A")

    def update_bounds(self, l, u):
        l, u = _f64(l).reshape(self.B, -1), _f64(u).reshape(self.B, -1)
        _chk(lib().mi_osqp_multi_batch_update_bounds(self._h, _dp(l), _dp(u)), "multi update_bounds")

    def update_A_bounds(self, Ax, l, u):
        Ax = _f64(Ax).reshape(self.B, -1)
        l, u = _f64(l).reshape(self.B, -1), _f64(u).reshape(self.B, -1)
        _chk(lib().mi_osqp_multi_batch_update_A_bounds(self._h, _ip(self._Ap), _ip(self._Ai), _dp(Ax), _dp(l), _dp(u)), "multi update_A_bounds")

    def warm_start_x(self, x):
        x = _f64(x).reshape(self.B, -1)
        _chk(lib().mi_osqp_multi_batch_warm_start_x(self._h, _dp(x)), "multi warm_start_x")


class QPSolver:
    """Python twin of the reference class QPSolver
    ([REF] /root/reference/src/osqp-wrapper.h:12-60): ctor(constraints, P),
    update(constraints), setWarmStart(x), solve() -> (exit_code, x).
    `constraints` = (l, A, u) like the reference's QPConstraints tuple."""

    def __init__(self, constraints, P, q=None, **settings):
        l, A, u = constraints
        self._b = BatchSolver(P, _csc(P).data, q, A, _csc(A).data, l, u, **settings)

    def update(self, constraints):
        l, A, u = constraints
        A = _csc(A)
        try:
            self._b.update_A(A.data, A_pattern=A)
            self._b.update_bounds(l, u)
        except MiOsqpError as e:           # the reference throws std::invalid_argument here
            raise ValueError(str(e)) from e

    def setWarmStart(self, x):
        self._b.warm_start_x(x)

    def solve(self):
        info = self._b.solve()[0]
        return int(info.exit_code), self._b.primal()[0]

    def info(self):
        return self._b.info()[0]

    def dual(self):
        return self._b.dual()[0]

    def stats(self):
        return self._b.stats()


def debug_host_kkt_solve(P, A, l, u, rhs, tri_waves=0, **settings):
    """Host-only: factor one QP's KKT and solve K sol = rhs twice -- by replaying
    the DEVICE schedules sequentially and by a plain CSC solve.  (No GPU.)
    tri_waves > 0: the dataflow form of the sweeps for that many waves (large single QPs)."""
    L = lib()
    P, A = _csc(P), _csc(A)
    n, m = A.shape[1], A.shape[0]
    s = default_settings(**settings)
    rhs = _f64(rhs)
    sol_s, sol_d = np.empty(n + m), np.empty(n + m)
    st = Stats()
    Pp, Pi, Px = _i64(P.indptr), _i64(P.indices), _f64(P.data)
    Ap, Ai, Ax = _i64(A.indptr), _i64(A.indices), _f64(A.data)
    l, u = _f64(l), _f64(u)
    rc = L.mi_osqp_debug_host_kkt_solve(n, m, _ip(Pp), _ip(Pi), _dp(Px), _ip(Ap), _ip(Ai), _dp(Ax), _dp(l), _dp(u),
                                        C.byref(s), 1 + int(tri_waves), _dp(rhs), _dp(sol_s), _dp(sol_d), C.byref(st))
    _chk(rc, "debug_host_kkt_solve")
    return sol_s, sol_d, st.as_dict()


def prefetch_analysis(P, A, B=1, device=-1, check=True):
    """Pattern analysis of a later setup of B QPs with the patterns of (P, A), computed now into the process-wide cache
    (host work only; blocking - call it from a spare thread).  Returns the C-ABI status."""
    L = lib()
    P, A = _csc(P), _csc(A)
    n, m = A.shape[1], A.shape[0]
    Pp, Pi, Ap, Ai = _i64(P.indptr), _i64(P.indices), _i64(A.indptr), _i64(A.indices)
    rc = L.mi_osqp_prefetch_analysis(B, n, m, _ip(Pp), _ip(Pi), _ip(Ap), _ip(Ai), device)
    if check:
        _chk(rc, "prefetch_analysis")
    return rc


def debug_host_block_factor(P, A, l, u, **settings):
    """Host-only: replay the device block refactorisation and compare with the host
    left-looking factor.  Returns (max rel diff L, max rel diff Dinv, counts dict)."""
    L = lib()
    P, A = _csc(P), _csc(A)
    n, m = A.shape[1], A.shape[0]
    s = default_settings(**settings)
    Pp, Pi, Px = _i64(P.indptr), _i64(P.indices), _f64(P.data)
    Ap, Ai, Ax = _i64(A.indptr), _i64(A.indices), _f64(A.data)
    l, u = _f64(l), _f64(u)
    dL, dD = C.c_double(), C.c_double()
    cnt = (C.c_int64 * 4)()
    rc = L.mi_osqp_debug_host_block_factor(n, m, _ip(Pp), _ip(Pi), _dp(Px), _ip(Ap), _ip(Ai), _dp(Ax), _dp(l), _dp(u),
                                           C.byref(s), C.byref(dL), C.byref(dD), cnt)
    _chk(rc, "debug_host_block_factor")
    return dL.value, dD.value, dict(blocks=cnt[0], triples=cnt[1], storage=cnt[2], levels=cnt[3])
