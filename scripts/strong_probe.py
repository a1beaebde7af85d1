"""Under-filled GPU (the strong-scaling shards of the 1024-QP batch: 128 / 256 QPs on one MI355X): solve time against the
tile shape.   python scripts/strong_probe.py"""
import importlib, os, subprocess, sys, time
import numpy as np
if len(sys.argv) > 1:
    B = int(sys.argv[1])
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    M = importlib.import_module("osqp-solver_amd")
    PR = importlib.import_module("osqp-solver_amd.problems")
    pr = PR.random_box_qp(1024)
    for k in ("Px", "Ax", "q", "l", "u"): pr[k] = pr[k][:B]
    s = M.BatchSolver(pr["P"], pr["Px"], pr["q"], pr["A"], pr["Ax"], pr["l"], pr["u"])
    s.solve()
    ts = []
    for _ in range(4):
        s.reset(); torch.cuda.synchronize()
        t = time.perf_counter(); info = s.solve(); ts.append(time.perf_counter() - t)
    st = s.stats()
    print(f"B={B} threads={st['threads_per_block']} tile={st['tile']} dense_tail={st['dense_tail_rows']}: {1e3 * min(ts):.2f} ms per solve, "
          f"{B / min(ts):.0f} QPs/s, iters max {max(i.iter for i in info)}, iterate launches (ms, count) {s.kernel_time()} refactor {s.refactor_time()}", flush=True)
    sys.exit(0)
for B in (128, 256, 512):
    for thr, tile in (("512", "1"), ("1024", "1"), ("512", "2"), ("1024", "2")):
        env = dict(os.environ, MI_OSQP_THREADS=thr, MI_OSQP_TILE=tile)
        subprocess.run([sys.executable, os.path.abspath(__file__), str(B)], env=env, timeout=300)
