"""developer probe: batch solve time of GOMP batches at one / two QPs per tile.   python scripts/tile_probe.py B D W"""
import importlib, os, subprocess, sys, time
if os.environ.get("TILE_PROBE_CHILD"):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    M = importlib.import_module("osqp-solver_amd")
    PR = importlib.import_module("osqp-solver_amd.problems")
    B, D, W = (int(v) for v in sys.argv[1:4])
    pr = PR.gomp_batch(B, D, W)
    s = M.BatchSolver(pr["P"], pr["Px"], None, pr["A"], pr["Ax"], pr["l"], pr["u"])
    s.warm_start_x(pr["warm"]); s.solve()
    ts = []
    for _ in range(5):
        s.reset(); s.warm_start_x(pr["warm"]); torch.cuda.synchronize()
        t = time.perf_counter(); info = s.solve(); ts.append(time.perf_counter() - t)
    t0 = time.perf_counter(); s.refactor_device(); tr = time.perf_counter() - t0
    st = s.stats()
    print(f"B={B} D={D} W={W} tile={st['tile']} threads={st['threads_per_block']} lds={st['lds_bytes']}: solve {1e3 * min(ts):.3f} ms ({max(i.iter for i in info)} it), refactor {1e3 * tr:.2f} ms", flush=True)
    sys.exit(0)
for t in ("2", "1", "2", "1"):
    subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:4], env=dict(os.environ, MI_OSQP_TILE=t, TILE_PROBE_CHILD="1"), timeout=300)
