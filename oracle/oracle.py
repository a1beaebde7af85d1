"""oracle/oracle.py -- ctypes binding of the CPU restatement (TEST INFRASTRUCTURE ONLY).

PARITY UNPINNED: see oracle/osqp_oracle.h.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this module; the product
(osqp-solver_amd/) never does.

Mirrors the method set of the reference's QPSolver
([REF] /root/reference/src/osqp-wrapper.h:16,33,45,51): ctor / update /
setWarmStart / solve.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_LIB_NATIVE = None

STATUS = {
    1: "kOptimal", 2: "kOptimalInaccurate", 3: "kPrimalInfeasibleInaccurate",
    4: "kDualInfeasibleInaccurate", -2: "kMaxIterations", -3: "kPrimalInfeasible",
    -4: "kDualInfeasible", -7: "kNonConvex", -10: "kUnknown",
}


class Settings(C.Structure):
    _fields_ = [
        ("rho", C.c_double), ("sigma", C.c_double), ("scaling", C.c_longlong),
        ("adaptive_rho", C.c_longlong), ("adaptive_rho_interval", C.c_longlong),
        ("adaptive_rho_tolerance", C.c_double), ("max_iter", C.c_longlong),
        ("eps_abs", C.c_double), ("eps_rel", C.c_double),
        ("eps_prim_inf", C.c_double), ("eps_dual_inf", C.c_double),
        ("alpha", C.c_double), ("scaled_termination", C.c_longlong),
        ("check_termination", C.c_longlong), ("warm_start", C.c_longlong),
    ]


class Info(C.Structure):
    _fields_ = [
        ("iter", C.c_longlong), ("status_val", C.c_longlong), ("obj_val", C.c_double),
        ("pri_res", C.c_double), ("dua_res", C.c_double), ("rho_updates", C.c_longlong),
        ("rho_estimate", C.c_double), ("rho", C.c_double), ("nnz_L", C.c_longlong),
    ]


def build(native=False):
    """Compile the C restatement (gcc).  native=True adds -march=native into a
    separate file so a library built in one container never runs foreign ISA."""
    out = os.path.join(_HERE, "_build")
    os.makedirs(out, exist_ok=True)
    name = "liboracle_osqp.native.so" if native else "liboracle_osqp.so"
    path = os.path.join(out, name)
    src = os.path.join(_HERE, "osqp_oracle.c")
    hdr = os.path.join(_HERE, "osqp_oracle.h")
    if os.path.exists(path) and not native and os.path.getmtime(path) >= max(os.path.getmtime(src), os.path.getmtime(hdr)):
        return path
    march = "native" if native else "x86-64-v3"
    cmd = ["gcc", "-O3", f"-march={march}", "-fPIC", "-fopenmp", "-std=c11", "-shared", "-o", path, src, "-lm"]
    subprocess.run(cmd, check=True)
    return path


def _bind(lib):
    ip = C.POINTER(C.c_longlong)
    dp = C.POINTER(C.c_double)
    lib.oq_default_settings.argtypes = [C.POINTER(Settings)]
    lib.oq_setup.restype = C.c_void_p
    lib.oq_setup.argtypes = [C.c_longlong, C.c_longlong, ip, ip, dp, dp, ip, ip, dp, dp, dp, C.POINTER(Settings), ip]
    lib.oq_setup_ordered.restype = C.c_void_p
    lib.oq_setup_ordered.argtypes = [C.c_longlong, C.c_longlong, ip, ip, dp, dp, ip, ip, dp, dp, dp, C.POINTER(Settings), ip, ip]
    lib.oq_solve.restype = C.c_longlong
    lib.oq_solve.argtypes = [C.c_void_p]
    lib.oq_get_solution.argtypes = [C.c_void_p, dp, dp]
    lib.oq_get_info.argtypes = [C.c_void_p, C.POINTER(Info)]
    lib.oq_update_A.restype = C.c_longlong
    lib.oq_update_A.argtypes = [C.c_void_p, ip, ip, dp]
    lib.oq_update_bounds.restype = C.c_longlong
    lib.oq_update_bounds.argtypes = [C.c_void_p, dp, dp]
    lib.oq_warm_start_x.restype = C.c_longlong
    lib.oq_warm_start_x.argtypes = [C.c_void_p, dp]
    lib.oq_cleanup.argtypes = [C.c_void_p]
    lib.oq_kkt_dim.restype = C.c_longlong
    lib.oq_kkt_dim.argtypes = [C.c_void_p]
    lib.oq_get_factor.argtypes = [C.c_void_p, ip, ip, ip, dp, dp]
    lib.oq_kkt_solve.argtypes = [C.c_void_p, dp, dp]
    lib.oq_batch_solve.restype = C.c_longlong
    lib.oq_batch_solve.argtypes = [C.c_longlong, C.c_longlong, C.c_longlong, ip, ip, dp, dp, ip, ip, dp, dp, dp,
                                   C.POINTER(Settings), C.c_longlong, dp, ip, ip, dp, dp]
    return lib


def lib(native=False):
    global _LIB, _LIB_NATIVE
    if native:
        if _LIB_NATIVE is None:
            try:
                _LIB_NATIVE = _bind(C.CDLL(build(native=True)))
            except Exception:
                _LIB_NATIVE = lib(False)
        return _LIB_NATIVE
    if _LIB is None:
        path = os.path.join(_HERE, "_build", "liboracle_osqp.so")
        try:
            path = build(False)
        except Exception:
            if not os.path.exists(path):
                raise
        _LIB = _bind(C.CDLL(path))
    return _LIB


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_longlong))


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def default_settings(**kw):
    s = Settings()
    lib().oq_default_settings(C.byref(s))
    for k, v in kw.items():
        if not hasattr(s, k):
            raise KeyError(k)
        setattr(s, k, v)
    return s


class OracleQPSolver:
    """CPU twin of the reference QPSolver.  P, A are scipy CSC matrices (any
    triangles of P; the upper one is used, as osqp-cpp does [EXT])."""

    def __init__(self, P, q, A, l, u, kkt_perm=None, **settings):
        import scipy.sparse as sp
        P = sp.csc_matrix(P); A = sp.csc_matrix(A)
        P.sort_indices(); A.sort_indices()
        self.n = A.shape[1]; self.m = A.shape[0]
        self._L = lib()
        self._Ap, self._Ai = _i64(A.indptr), _i64(A.indices)
        Pp, Pi, Px = _i64(P.indptr), _i64(P.indices), _f64(P.data)
        Ax = _f64(A.data)
        qv = None if q is None else _f64(q)
        lv, uv = _f64(l), _f64(u)
        self.settings = default_settings(**settings)
        err = C.c_longlong(0)
        if kkt_perm is None:
            self._h = self._L.oq_setup(self.n, self.m, _ip(Pp), _ip(Pi), _dp(Px), _dp(qv),
                                       _ip(self._Ap), _ip(self._Ai), _dp(Ax), _dp(lv), _dp(uv),
                                       C.byref(self.settings), C.byref(err))
        else:       # a caller-supplied elimination order (the exact minimum degree is quadratic in the fill)
            pm = _i64(kkt_perm)
            self._h = self._L.oq_setup_ordered(self.n, self.m, _ip(Pp), _ip(Pi), _dp(Px), _dp(qv),
                                               _ip(self._Ap), _ip(self._Ai), _dp(Ax), _dp(lv), _dp(uv),
                                               C.byref(self.settings), _ip(pm), C.byref(err))
        if not self._h:
            raise ValueError(f"oracle setup failed, err={err.value}")

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.oq_cleanup(self._h); self._h = None

    def solve(self):
        st = self._L.oq_solve(self._h)
        x = np.empty(self.n); y = np.empty(self.m)
        self._L.oq_get_solution(self._h, _dp(x), _dp(y))
        self.y = y
        return int(st), x

    def info(self):
        i = Info(); self._L.oq_get_info(self._h, C.byref(i)); return i

    def update(self, l, A, u):
        import scipy.sparse as sp
        A = sp.csc_matrix(A); A.sort_indices()
        Ap, Ai, Ax = _i64(A.indptr), _i64(A.indices), _f64(A.data)
        if len(Ap) != len(self._Ap) or len(Ai) != len(self._Ai):
            raise ValueError("sparsity pattern changed")
        if self._L.oq_update_A(self._h, _ip(Ap), _ip(Ai), _dp(Ax)) != 0:
            raise ValueError("update_A failed (pattern changed or refactor failed)")
        lv, uv = _f64(l), _f64(u)
        if self._L.oq_update_bounds(self._h, _dp(lv), _dp(uv)) != 0:
            raise ValueError("lower bound must be <= upper bound")

    def update_bounds_only(self, l, u):
        lv, uv = _f64(l), _f64(u)
        if self._L.oq_update_bounds(self._h, _dp(lv), _dp(uv)) != 0:
            raise ValueError("lower bound must be <= upper bound")

    def set_warm_start(self, x):
        xv = _f64(x); self._L.oq_warm_start_x(self._h, _dp(xv))

    def factor(self):
        N = int(self._L.oq_kkt_dim(self._h))
        nnz = int(self.info().nnz_L)
        perm = np.empty(N, np.int64); Lp = np.empty(N + 1, np.int64)
        Li = np.empty(nnz, np.int64); Lx = np.empty(nnz); Dinv = np.empty(N)
        self._L.oq_get_factor(self._h, _ip(perm), _ip(Lp), _ip(Li), _dp(Lx), _dp(Dinv))
        return perm, Lp, Li, Lx, Dinv

    def kkt_solve(self, rhs):
        rhs = _f64(rhs); sol = np.empty_like(rhs)
        self._L.oq_kkt_solve(self._h, _dp(rhs), _dp(sol)); return sol


def batch_solve(P_pattern, Px, q, A_pattern, Ax, l, u, threads=1, native=False, **settings):
    """B independent QPs, one shared pattern (scipy CSC matrices carry the
    pattern; Px[B,nnzP], Ax[B,nnzA], q[B,n], l/u[B,m]).  Returns dict."""
    import scipy.sparse as sp
    L = lib(native)
    P = sp.csc_matrix(P_pattern); A = sp.csc_matrix(A_pattern)
    n, m = A.shape[1], A.shape[0]
    Px, Ax, l, u = _f64(Px), _f64(Ax), _f64(l), _f64(u)
    B = Ax.shape[0]
    qv = None if q is None else _f64(q)
    Pp, Pi = _i64(P.indptr), _i64(P.indices)
    Ap, Ai = _i64(A.indptr), _i64(A.indices)
    s = default_settings(**settings)
    x = np.empty((B, n)); status = np.empty(B, np.int64); iters = np.empty(B, np.int64)
    ts, tv = C.c_double(0), C.c_double(0)
    fail = L.oq_batch_solve(B, n, m, _ip(Pp), _ip(Pi), _dp(Px), _dp(qv), _ip(Ap), _ip(Ai), _dp(Ax),
                            _dp(l), _dp(u), C.byref(s), threads, _dp(x), _ip(status), _ip(iters),
                            C.byref(ts), C.byref(tv))
    return dict(x=x, status=status, iters=iters, setup_s=ts.value, solve_s=tv.value, failed=int(fail))
