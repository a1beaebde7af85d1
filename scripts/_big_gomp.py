import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import osqp_solver_amd as M
from osqp_solver_amd import problems as PR
from oracle import oracle as O
W = int(os.environ.get("W", "802"))
P, (l, A, u), _ = PR.gomp_qp(6, W, np.zeros(6), np.ones(6))
t = time.perf_counter(); s = M.QPSolver((l, A, u), P); t1 = time.perf_counter() - t
st = s.stats()
print({k: st[k] for k in ("N", "nnz_L", "fwd_levels", "bwd_levels", "fwd_slots", "tile", "dense_tail_rows", "lds_bytes")})
t = time.perf_counter(); code, x = s.solve(); t2 = time.perf_counter() - t
it1 = s.info().iter
t = time.perf_counter(); code2, x2 = s.solve(); t3 = time.perf_counter() - t
print(f"GPU: ctor {t1*1e3:.1f} ms, solve {t2*1e3:.2f} ms ({it1} it), warm re-solve {t3*1e3:.2f} ms ({s.info().iter} it)")
t = time.perf_counter(); o = O.OracleQPSolver(P, None, A, l, u); t4 = time.perf_counter() - t
t = time.perf_counter(); sto, xo = o.solve(); t5 = time.perf_counter() - t
print(f"oracle: setup {t4*1e3:.1f} ms, solve {t5*1e3:.2f} ms ({o.info().iter} it); max|dx| {np.abs(x - xo).max():.2e}")
