"""First-contact GPU check (developer script, not a test): small config-3 batch
vs the oracle, the two ops, then a timing of the full 1024 batch."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import osqp_solver_amd as M
from osqp_solver_amd import problems as PR
from oracle import oracle as O

def run(B, tile=None, full_oracle=8, **settings):
    if tile: os.environ["MI_OSQP_TILE"] = str(tile)
    pr = PR.random_box_qp(B)
    t = time.time()
    s = M.BatchSolver(pr["P"], pr["Px"], pr["q"], pr["A"], pr["Ax"], pr["l"], pr["u"], **settings)
    st = s.stats()
    print(f"B={B} setup {time.time()-t:.2f}s tile={st['tile']} tiles={st['n_tiles']} lds={st['lds_bytes']} nnzL={st['nnz_L']} "
          f"host={st['setup_seconds_host']:.2f} factor={st['setup_seconds_factor']:.2f} upload={st['setup_seconds_upload']:.2f}", flush=True)
    # ops
    nb = min(B, full_oracle)
    n, m = pr["n"], pr["m"]
    rng = np.random.default_rng(1)
    rhs = torch.tensor(rng.standard_normal((B, n + m)), device="cuda")
    sol = torch.empty_like(rhs)
    s.kkt_solve_device(rhs, sol)
    for b in range(nb):
        P, A = PR.qp_matrices(pr, b)
        o = O.OracleQPSolver(P, pr["q"][b], A, pr["l"][b], pr["u"][b], **settings)
        ref = o.kkt_solve(rhs[b].cpu().numpy())
        err = np.max(np.abs(ref - sol[b].cpu().numpy())) / np.max(np.abs(ref))
        if b < 2 or err > 1e-9: print("  kkt_solve rel err", b, err)
    t = time.time(); info = s.solve(); dt = time.time() - t
    ls = s.last_solve_stats()
    its = np.array([i.iter for i in info]); stv = np.array([i.status_val for i in info])
    print(f"  solve {dt*1e3:.1f} ms  iters mean {its.mean():.1f} max {its.max()} status {np.unique(stv, return_counts=True)} {ls}", flush=True)
    x = s.primal()
    worst = 0
    for b in range(nb):
        P, A = PR.qp_matrices(pr, b)
        o = O.OracleQPSolver(P, pr["q"][b], A, pr["l"][b], pr["u"][b], **settings)
        so, xo = o.solve(); io = o.info()
        err = np.max(np.abs(xo - x[b]))
        worst = max(worst, err)
        if b < 3 or err > 1e-6 or io.iter != info[b].iter:
            print(f"  qp{b}: oracle st {so} it {io.iter} rho_upd {io.rho_updates} | gpu st {info[b].status_val} it {info[b].iter} rho_upd {info[b].rho_updates} | max|dx| {err:.2e} pri {io.pri_res:.3e}/{info[b].pri_res:.3e}")
    print("  worst primal err", worst, flush=True)
    # repeat timing with reset
    for k in range(3):
        s.reset(); torch.cuda.synchronize()
        t = time.time(); s.solve(); dt = time.time() - t
        ls = s.last_solve_stats()
        print(f"  re-solve {dt*1e3:.1f} ms  {B/dt:.0f} QPs/s  {ls}", flush=True)
    s.close()

if __name__ == "__main__":
    run(8)
    run(8, eps_abs=1e-8, eps_rel=1e-8)
    run(64, tile=4)
    run(1024, full_oracle=4)
