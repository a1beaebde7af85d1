"""developer script: per-op timings on the headline config (B=1024, n=512, m=1024)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import osqp_solver_amd as M
from osqp_solver_amd import problems as PR

B = int(os.environ.get("B", "1024"))
pr = PR.random_box_qp(B)
def mk(**kw):
    return M.BatchSolver(pr["P"], pr["Px"], pr["q"], pr["A"], pr["Ax"], pr["l"], pr["u"], **kw)
s = mk()
st = s.stats()
print({k: st[k] for k in ("tile", "n_tiles", "nnz_L", "fwd_levels", "bwd_levels", "fwd_slots", "bwd_slots", "chk_slots", "dense_tail_rows", "dense_tail_slots", "lds_bytes", "threads_per_block")})
n, m = pr["n"], pr["m"]
rhs = torch.randn(B, n + m, dtype=torch.float64, device="cuda"); sol = torch.empty_like(rhs)
x = torch.randn(B, n, dtype=torch.float64, device="cuda"); y = torch.randn(B, m, dtype=torch.float64, device="cuda")
Px = torch.empty_like(x); Aty = torch.empty_like(x); Ax = torch.empty_like(y)
def timeit(f, reps=20):
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps
t = timeit(lambda: s.kkt_solve_device(rhs, sol))
bytes_solve = (st["fwd_slots"] + st["bwd_slots"] + st["dense_tail_slots"]) * 8 * B
print(f"kkt_solve op: {t*1e3:.3f} ms  -> {bytes_solve/t/1e9:.0f} GB/s of streamed (padded) factor values; SURVEY 8(d) algorithmic {2*8*st['nnz_L']*B/t/1e9:.0f} GB/s")
t = timeit(lambda: s.spmv_device(x, y, Px, Aty, Ax))
alg = (2 * (8 * st["nnz_A"] + 8 * n + 8 * m) + 8 * st["nnz_P_triu"] + 16 * n) * B
print(f"spmv op: {t*1e6:.1f} us -> algorithmic {alg/t/1e9:.0f} GB/s, padded values {st['chk_slots']*8*B/t/1e9:.0f} GB/s")
t = timeit(lambda: s.refactor_device(), reps=5)
print(f"refactor_device (all {B} QPs): {t*1e3:.2f} ms")
s.close()
# fixed 100 iterations, no checks, no rho adaptation
s = mk(max_iter=100, check_termination=0, adaptive_rho=0)
s.solve(); 
t0 = time.perf_counter(); s.reset(); s.solve(); torch.cuda.synchronize(); t = time.perf_counter() - t0
ls = s.last_solve_stats()
print(f"100 iterations, no checks: {t*1e3:.1f} ms wall, device {ls['device_s']*1e3:.1f} ms -> {ls['device_s']*1e3/100:.3f} ms/iteration")
s.close()
s = mk(max_iter=100, check_termination=25, adaptive_rho=0, eps_abs=1e-14, eps_rel=1e-14)
s.solve(); s.reset(); s.solve(); ls = s.last_solve_stats()
print(f"100 iterations + 4 checks: device {ls['device_s']*1e3:.1f} ms")
