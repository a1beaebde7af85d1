"""512 against 1024 threads per tile on the GOMP batches (configs 2 and 4).   python scripts/threads_probe.py"""
import importlib, os, subprocess, sys, time
if len(sys.argv) > 1:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    M = importlib.import_module("osqp-solver_amd")
    PR = importlib.import_module("osqp-solver_amd.problems")
    Bq, D, W = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    pr = PR.gomp_batch(Bq, D, W)
    s = M.BatchSolver(pr["P"], pr["Px"], None, pr["A"], pr["Ax"], pr["l"], pr["u"])
    s.warm_start_x(pr["warm"]); s.solve()
    ts = []
    for _ in range(5):
        s.reset(); s.warm_start_x(pr["warm"]); torch.cuda.synchronize()
        t = time.perf_counter(); info = s.solve(); ts.append(time.perf_counter() - t)
    st = s.stats()
    print(f"B={Bq} D={D} W={W} threads={st['threads_per_block']} tile={st['tile']}: {1e3 * min(ts):.3f} ms per solve, iters {max(i.iter for i in info)}", flush=True)
    sys.exit(0)
for cfg in (("1", "6", "50"), ("256", "7", "100"), ("64", "7", "100"), ("1024", "7", "100")):
    for thr in ("512", "1024"):
        subprocess.run([sys.executable, os.path.abspath(__file__), *cfg], env=dict(os.environ, MI_OSQP_THREADS=thr), timeout=300)
