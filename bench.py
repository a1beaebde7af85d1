#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric on BASELINE config 3.

step     = one pass of the hot path over one batch: reset to the post-setup state
           (cold start), refresh the bounds from HBM-resident tensors, solve all
           1024 box-QPs (n=512, m=1024) of this rank to OSQP's default tolerance,
           leave x in HBM; with N>1 ranks additionally all_gather the solutions
           (the one collective the path has).
value    = QPs/s over all ranks (weak scaling: 1024 QPs per GPU), inputs resident
           in HBM when the timed region starts; setup (host analysis + first
           factorisation + upload) is outside and reported separately.
roofline = admm_kernel: algorithmic bytes (SURVEY.md section 8(d) formulas x the QP
           iterations the launches processed) / summed launch durations measured with
           HIP events on the launch stream.
cpu_baseline = the oracle (CPU restatement, kind "port") on a bounded sample of the
           same workload on this box's host cores.

Launch:  python bench.py [--gpus N --steps K --warmup W]
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def algorithmic_bytes(st, iters, n_checks_per_qp):
    """values per QP, index arrays once per distinct pattern."""
    n, m, N, nnzL = st["n"], st["m"], st["N"], st["nnz_L"]
    nnzA, nnzP = st["nnz_A"], st["nnz_P_triu"]
    tri = 2 * 8 * nnzL + 8 * N + 2 * 2 * 8 * N           # L twice, D^-1, rhs read+write per sweep
    vec = 8 * (5 * n + 9 * m)                             # E6 + E8-E10
    per_iter = tri + vec
    spmv = (8 * nnzA + 8 * n + 8 * m) * 2 + (8 * nnzP + 16 * n) + 8 * (3 * n + 3 * m)
    pattern_per_iter = 2 * (4 * nnzL + 4 * (N + 1))
    total_iters = int(np.sum(iters))
    return dict(per_qp_iter=per_iter, per_qp_check=spmv,
                total=per_iter * total_iters + spmv * int(np.sum(n_checks_per_qp)) + pattern_per_iter * int(np.max(iters)),
                tri_pair=tri)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1024, help="QPs per GPU (BASELINE config 3: 1024)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist_on = world > 1
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU solve path)")
    torch.cuda.set_device(local_rank)
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    import osqp_solver_amd as M
    from osqp_solver_amd import problems as PR
    from osqp_solver_amd.sharding import gather_solutions

    B = args.batch
    # weak scaling: every rank owns B QPs with its own values (same pattern)
    pr = PR.random_box_qp(B, value_seed=1000 + rank * B)
    t0 = time.time()
    solver = M.BatchSolver(pr["P"], pr["Px"], pr["q"], pr["A"], pr["Ax"], pr["l"], pr["u"], device=local_rank)
    setup_s = time.time() - t0
    st = solver.stats()
    dev = torch.device("cuda", local_rank)
    d_l = torch.tensor(pr["l"], device=dev); d_u = torch.tensor(pr["u"], device=dev)
    d_x = torch.empty(B, pr["n"], dtype=torch.float64, device=dev)
    d_status = torch.empty(B, dtype=torch.int32, device=dev); d_iters = torch.empty_like(d_status)

    def step():
        solver.reset()
        solver.update_bounds_device(d_l, d_u)
        solver.solve_device(d_x, d_status, d_iters)
        if dist_on:
            return gather_solutions(d_x)
        return d_x

    def fence():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    solver.kernel_time()                       # clear the HIP-event accumulators
    fence()
    t0 = time.perf_counter()
    dev_s = ref_s = cmp_s = 0.0
    for _ in range(args.steps):
        step()
        ls = solver.last_solve_stats()
        dev_s += ls["device_s"]; ref_s += ls["refactor_s"]; cmp_s += ls["compact_s"]
    fence()
    elapsed = time.perf_counter() - t0
    if dist_on:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    avg_ms, launches = solver.kernel_time()
    iters = d_iters.cpu().numpy().astype(np.int64)
    status = d_status.cpu().numpy()
    ls = solver.last_solve_stats()
    total_qps = B * world * args.steps
    value = total_qps / elapsed
    iters_per_s = float(iters.sum()) * world * args.steps / elapsed

    out = {
        "metric": "QPs/sec on a 1024-QP batch of box-QPs (n=512, m=1024), OSQP defaults; ADMM iters/sec alongside",
        "value": value, "unit": "QPs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "BASELINE config 3: batch of 1024 random box-QPs n=512 m=1024 per GPU, shared pattern, "
                               "A=[I;G] 8 nnz/row, strictly convex banded P; eps_abs=eps_rel=1e-3, adaptive rho interval 100",
                   "qps_per_gpu": B, "tile": st["tile"], "n_tiles": st["n_tiles"], "nnz_L": st["nnz_L"],
                   "fwd_levels": st["fwd_levels"], "bwd_levels": st["bwd_levels"], "dense_tail_rows": st["dense_tail_rows"]},
        "admm_iters_per_sec": iters_per_s,
        "iters_mean": float(iters.mean()), "iters_max": int(iters.max()),
        "all_solved": bool(np.all(status == 1)),
        "setup_seconds": setup_s,
        "step_breakdown_ms": {"device_iterate": 1e3 * dev_s / args.steps, "device_refactor": 1e3 * ref_s / args.steps, "compaction": 1e3 * cmp_s / args.steps,
                              "refactors_per_step": ls["refactors"], "launches_per_step": ls["launches"]},
    }
    if rank == 0:
        ab = algorithmic_bytes(st, iters, iters // 25)
        launches_per_step = max(1, launches // max(1, args.steps))
        kernel_s_per_step = avg_ms * 1e-3 * launches_per_step
        achieved = ab["total"] / kernel_s_per_step / 1e9 if kernel_s_per_step > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("bytes_per_launch")
            except Exception:
                traffic = None
        out["roofline"] = {"bound": "hbm", "kernel": "iterate_kernel<%d,512>" % st["tile"], "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                           "frac": achieved / 8000.0, "traffic": traffic,
                           "algorithmic_bytes_per_launch": ab["total"] / launches_per_step,
                           "avg_launch_ms": avg_ms, "launches_per_step": launches_per_step,
                           "bytes_per_qp_iteration": ab["per_qp_iter"]}
        # what the kernel actually streams per QP-iteration: the padded forward / backward step streams, the inverted
        # Schur complement of the dense tail (read ONCE where SURVEY 8(d) counts its triangle of L twice), D^-1, vectors
        streamed = 8 * (st["fwd_slots"] + st["bwd_slots"] + st["dense_tail_slots"]) + 8 * st["N"] + 8 * (5 * st["n"] + 9 * st["m"])
        out["roofline"]["streamed_bytes_per_qp_iteration"] = streamed
        out["roofline"]["streamed_frac"] = out["roofline"]["frac"] * streamed / ab["per_qp_iter"]
        out["roofline"]["note"] = ("achieved/frac use the algorithmic bytes of SURVEY 8(d) (two triangular sweeps over L); with the dense tail the "
                                   "kernel reads fewer bytes than that (streamed_*), so traffic < algorithmic")
        if not args.no_cpu_baseline and world == 1:
            from oracle import oracle as O
            cores = M.host_cores()             # min(affinity, cgroup quota): the box shows 256 CPUs, grants ~16
            sample = int(min(B, max(32, 8 * cores)))
            r = O.batch_solve(pr["P"], pr["Px"][:sample], pr["q"][:sample], pr["A"], pr["Ax"][:sample],
                              pr["l"][:sample], pr["u"][:sample], threads=cores, native=True)
            same = bool(np.array_equal(r["iters"], iters[:sample]))
            err = float(np.max(np.abs(r["x"] - d_x[:sample].cpu().numpy())))
            r1 = O.batch_solve(pr["P"], pr["Px"][:8], pr["q"][:8], pr["A"], pr["Ax"][:8], pr["l"][:8], pr["u"][:8],
                               threads=1, native=True)
            out["cpu_baseline"] = {"value": sample / r["solve_s"], "unit": "QPs/s", "cores": cores, "kind": "port",
                                   "sample": f"first {sample} QPs of the same batch, oracle solve phase on {cores} threads "
                                             f"(setup {r['setup_s']:.2f}s excluded, as for the GPU); single-thread rate "
                                             f"{8 / r1['solve_s']:.1f} QPs/s on 8 QPs",
                                   "single_thread_qps": 8 / r1["solve_s"], "setup_seconds_sample": r["setup_s"],
                                   "gpu_vs_cpu_max_abs_x_diff": err, "same_iteration_counts": same}
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
