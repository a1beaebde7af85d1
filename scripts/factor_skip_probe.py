"""developer script (diagnostic build only): where the refactorisation of a GOMP batch spends its time - the same update with
parts of factor_kernel switched off (MI_OSQP_FACTOR_SKIP bits: 1 rank-1 updates, 2 general updates, 4 diagonal blocks,
8 triangular solves, 16 scatter into the solve streams; results are garbage, only the time means something).
   MI_OSQP_LIBRARY=osqp-solver_amd/libmi_osqp_debug.so python scripts/factor_skip_probe.py [B D W]"""
import importlib, os, subprocess, sys, time
if len(sys.argv) > 4:
    import numpy as np
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    M = importlib.import_module("osqp-solver_amd")
    PR = importlib.import_module("osqp-solver_amd.problems")
    B, D, W = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    if D == 0:                   # D = 0: the random QPs of the headline batch (config 3), W ignored
        pr = PR.random_box_qp(1024)
        for k in ("Px", "Ax", "q", "l", "u"): pr[k] = pr[k][:B]
        s = M.BatchSolver(pr["P"], pr["Px"], pr["q"], pr["A"], pr["Ax"], pr["l"], pr["u"])
    else:
        pr = PR.gomp_batch(B, D, W)
        s = M.BatchSolver(pr["P"], pr["Px"], None, pr["A"], pr["Ax"], pr["l"], pr["u"])
    ts = []
    for k in range(5):
        torch.cuda.synchronize(); t = time.perf_counter()
        s.refactor_device()
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    print(f"skip={os.environ.get('MI_OSQP_FACTOR_SKIP', '0'):>2}: refactor_device of {B} QPs (D={D}, W={W}): {1e3 * min(ts):.3f} ms", flush=True)
    sys.exit(0)
args = sys.argv[1:4] if len(sys.argv) > 3 else ["256", "7", "100"]
for skip in ("0", "3", "4", "8", "16", "7", "15", "31"):
    subprocess.run([sys.executable, os.path.abspath(__file__)] + args + ["x"], env=dict(os.environ, MI_OSQP_FACTOR_SKIP=skip), timeout=300)
