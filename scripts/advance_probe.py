"""developer probe: the round trip of mi_osqp_batch_advance + poll for one handle alone and for ten handles driven from ten
host threads at once (the stage threads of ContinuousGOMPSolver)."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import osqp_solver_amd as M
from osqp_solver_amd import problems as PR

B = 256
Ws = [100, 90, 80, 70, 60, 50, 40, 30, 20, 10]
hs = []
for W in Ws:
    pr = PR.gomp_batch(1, 3, W)
    rep = lambda a: np.repeat(a, B, axis=0)
    s = M.BatchSolver(pr["P"], rep(pr["Px"]), None, pr["A"], rep(pr["Ax"]), rep(pr["l"]), rep(pr["u"]), max_iter=100000, eps_abs=1e-30, eps_rel=1e-30, adaptive_rho=0)
    hs.append((W, s))
nrun = int(os.environ.get("NRUN", "4"))
for W, s in hs:
    s.solve_begin_some(list(range(nrun)))          # never converges at eps 1e-30: keeps iterating

def loop(s, reps, out):
    t = time.perf_counter()
    for _ in range(reps):
        s.advance(1); s.poll(True)
    out.append((time.perf_counter() - t) / reps)

for W, s in hs:
    o = []; loop(s, 20, o); o = []; loop(s, 100, o)
    print(f"alone      W={W:3d}: {1e3*o[0]:.3f} ms per advance+poll", flush=True)
outs = [[] for _ in hs]
th = [threading.Thread(target=loop, args=(s, 100, outs[i])) for i, (W, s) in enumerate(hs)]
t = time.perf_counter(); [x.start() for x in th]; [x.join() for x in th]; tot = time.perf_counter() - t
for (W, s), o in zip(hs, outs):
    print(f"10 threads W={W:3d}: {1e3*o[0]:.3f} ms per advance+poll", flush=True)
print(f"10 threads: 100 advances each in {tot:.3f} s")
# one thread, round robin: enqueue all, then poll all
t = time.perf_counter()
for _ in range(100):
    for W, s in hs: s.advance(1)
    for W, s in hs: s.poll(True)
tot = time.perf_counter() - t
print(f"1 thread, 10 handles round robin: {1e3*tot/100:.3f} ms per round")
