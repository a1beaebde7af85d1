"""developer script: where does a KKT solve spend its time?  Runs the traced twin of the kkt_solve op on the
headline batch (or B QPs) and prints, per sweep, the barrier-to-barrier time split by phase kind and by the
number of steps of the busiest wave, plus the critical-wave work vs. barrier-wait decomposition."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import osqp_solver_amd as M
from osqp_solver_amd import problems as PR

B = int(os.environ.get("B", "1024"))
os.environ.setdefault("MI_OSQP_TILE", "2"); os.environ.setdefault("MI_OSQP_THREADS", "512")      # the shape the traced twin exists for
if os.environ.get("GOMP"):                                 # GOMP=D,W: a batch of joint-space GOMP QPs instead of the headline batch
    D, W = (int(v) for v in os.environ["GOMP"].split(","))
    pr = PR.gomp_batch(B, D, W)
    s = M.BatchSolver(pr["P"], pr["Px"], None, pr["A"], pr["Ax"], pr["l"], pr["u"])
else:
    pr = PR.random_box_qp(B)
    s = M.BatchSolver(pr["P"], pr["Px"], pr["q"], pr["A"], pr["Ax"], pr["l"], pr["u"])
n, m = pr["n"], pr["m"]
rhs = torch.randn(B, n + m, dtype=torch.float64, device="cuda"); sol = torch.empty_like(rhs)
for _ in range(3):
    tr, ftab, btab, (fp, bp, nw, words) = s.debug_trace_kkt_solve(rhs, sol)
sol2 = torch.empty_like(rhs); s.kkt_solve_device(rhs, sol2)
print("trace solve == op solve:", bool(torch.equal(sol, sol2)))
for ti in range(2):
    t = tr[ti]
    dt_clk = np.uint32(t[2] - t[0]); dt_real = np.uint32(t[3] - t[1])
    if dt_real == 0: continue                              # (a batch of one tile has no second traced tile)
    mhz = float(dt_clk) / (float(dt_real) / 100.0)      # s_memrealtime ticks at 100 MHz
    print(f"tile sel {ti}: total {float(dt_real)/100.0:.1f} us, memtime clock {mhz:.0f} MHz")
    print(f"  between the sweeps (D^-1 scaling, dense tail product): {float(np.uint32(t[5] - t[4])) / mhz:.1f} us")
    tw = t[8: 8 + 4 * nw].reshape(2, nw, 2).astype(np.int64)
    off = 8 + 4 * nw
    for si, (name, tab, P) in enumerate((("fwd", ftab, fp), ("bwd", btab, bp))):
        print(f"  {name}: per wave waiting for ring data (us): {[round(float(c) / mhz, 1) for c in tw[si, :, 0]]} steps {tw[si, :, 1].tolist()}")
        st = t[off: off + P * nw * 2].reshape(P, nw, 2).astype(np.int64); off += P * nw * 2
        before, after = st[:, :, 0], st[:, :, 1]
        kind = tab[:, 0]
        begin = tab[:, 1::4][:, :nw].astype(np.int64); end = tab[:, 2::4][:, :nw].astype(np.int64)
        has = tab[:, 4::4][:, :nw]
        nsteps = end - begin
        start = np.empty_like(before); start[0] = before[0].min(); start[1:] = after[:-1]
        # barrier-to-barrier duration seen by wave 0
        release = after.max(axis=1)
        dur = np.diff(np.concatenate([[before[0].min()], release]))
        work = before - start                                 # per wave: phase entry -> arrival at barrier
        crit = work.max(axis=1)
        us = lambda c: c / mhz
        print(f"  {name}: {P} phases, total {us(dur.sum()):.1f} us; critical-wave work {us(crit.sum()):.1f} us, "
              f"barrier release after last arrival {us((release - before.max(axis=1)).sum()):.1f} us")
        for k in (0, 1):
            sel = kind == k
            if sel.any():
                print(f"    kind {k}: {int(sel.sum())} phases, {us(dur[sel].sum()):.1f} us, mean {us(dur[sel].mean()):.2f} us; "
                      f"steps of busiest wave mean {nsteps[sel].max(axis=1).mean():.1f}, all waves mean {nsteps[sel].mean():.2f}")
        a = kind == 0
        mx = nsteps.max(axis=1)
        for lo, hi in ((0, 1), (1, 2), (2, 3), (3, 5), (5, 9), (9, 16), (16, 10 ** 6)):
            sel = a & (mx >= lo) & (mx < hi)
            if sel.any():
                print(f"      A phases with busiest wave {lo}..{hi-1} steps: {int(sel.sum())} phases, mean {us(dur[sel].mean()):.2f} us, "
                      f"total {us(dur[sel].sum()):.1f} us")
        if os.environ.get("DUMP"):
            for p in range(P):
                print(f"      p{p} kind {kind[p]} dur {us(dur[p]):.2f} steps {nsteps[p].tolist()} work {[round(us(w), 2) for w in work[p]]}")
s.close()
