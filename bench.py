#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric on BASELINE config 3.

step     = one pass of the hot path over one batch: reset to the post-setup state
           (cold start), refresh the bounds from HBM-resident tensors, solve every
           box-QP (n=512, m=1024) of this rank to OSQP's default tolerance, leave x in
           HBM; with N>1 ranks additionally all_gather the solutions (the one
           collective the path has; counts are static and exchanged outside the step).
value    = QPs/s over all ranks, inputs resident in HBM when the timed region starts;
           setup (host analysis + first factorisation + upload) is outside and reported
           separately.  --scaling weak (default): 1024 QPs per GPU; --scaling strong: the
           1024-QP batch of the metric split over the ranks (sharding.shard_range).  For
           N>1 the other mode is measured in the same run and reported under "strong" /
           "weak" next to the primary numbers.
roofline = iterate_kernel: the MINIMAL bytes of the algorithm the kernel implements
           (L before the dense tail twice, the inverted Schur complement once, vectors;
           unpadded) x the QP-iterations of the launches / the launch durations measured
           with HIP events on the launch stream.  The SURVEY 8(d) figure (two sweeps over
           all of L) stays as frac_survey_8d.  kernels[] holds the refactorisation kernels.
cpu_baseline = the oracle (CPU restatement, kind "port") on a bounded sample of the
           same workload on this box's host cores.
secondary = BASELINE configs 2, 4, 5 and the GOMP obstacle scene (continuous driver) measured outside the headline
           timing (rank 0, N=1).
value_pcie_inclusive = the same step through the host-pointer boundary (H2D bounds, D2H solutions); roofline carries
           measured_copy_GBps / measured_triad_GBps of this box next to the nominal peak.

Launch:  python bench.py [--gpus N --steps K --warmup W]       (N>1: starts the N ranks itself)
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HEADLINE_B = 1024


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=HEADLINE_B, help="QPs of the headline batch (BASELINE config 3: 1024)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: --batch QPs per GPU; strong: --batch QPs in total, split over the GPUs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip BASELINE configs 2, 4, 5")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` run as a plain command: start N ranks (one per GPU) BEFORE anything touches the GPU,
    relay their output, exit with their code.  (Never re-exec a process that has initialised HIP.)"""
    import torch
    have = torch.cuda.device_count()              # (counting devices does not initialise the GPU on this image)
    if have < args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} requested but only {have} GPU(s) are visible; refusing to report a "
                         f"{have}-GPU number as the {args.gpus}-GPU point\n")
        sys.exit(2)
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), MASTER_ADDR="127.0.0.1")
    res = subprocess.run(cmd, env=env)
    sys.exit(res.returncode)


def algorithmic_bytes_8d(st):
    """values per QP (index arrays once per pattern); two sweeps over ALL of L."""
    n, m, N, nnzL = st["n"], st["m"], st["N"], st["nnz_L"]
    tri = 2 * 8 * nnzL + 8 * N + 2 * 2 * 8 * N           # L twice, D^-1, rhs read+write per sweep
    vec = 8 * (5 * n + 9 * m)                             # E6 + E8-E10
    return tri + vec


def minimal_bytes(st):
    """Minimal bytes per QP-iteration of the algorithm the kernel IMPLEMENTS (unpadded): the factor entries before the
    dense tail twice (forward + backward sweep), the inverted Schur complement of the k tail rows once (k(k+1)/2
    values incl. its diagonal), D^-1 and the rhs traffic of the sweeps (40 N), the vector step 8(5n+9m)."""
    n, m, N, k = st["n"], st["m"], st["N"], st["dense_tail_rows"]
    return 8 * (2 * st["nnz_L_before_tail"] + k * (k + 1) // 2) + 40 * N + 8 * (5 * n + 9 * m)


def refactor_bytes(st):
    """Algorithmic bytes per refactored QP of the two E13 kernels (values only; tables are shared and L2 resident).
    factor_kernel: reads the KKT values (P, A, rho: nnz_KKT), writes the factor before the tail in BOTH sweep orders plus
    D^-1 and the KKT part of the k x k tail (lower triangle).  tail kernels (tail_assemble_kernel + tail_kernel): read that
    triangle (the factor entries they gather are counted with factor_kernel), write the k(k+1)/2 values of S^-1."""
    k = st["dense_tail_rows"]
    tri = k * (k + 1) // 2
    return {"factor_kernel": 8 * (st["nnz_KKT"] + 2 * st["nnz_L_before_tail"] + st["N"] + tri),
            "tail_kernels": 8 * (tri + tri)}


def load_traffic():
    """PMC traffic per launch of the committed profile of this round (profiles/hbm_traffic.json)."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
    except Exception:
        return {}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)                     # does not return
    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with --nproc-per-node {args.gpus}")
    dist_on = world > 1
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU solve path)")
    torch.cuda.set_device(local_rank)
    dist = None
    if dist_on or os.environ.get("MI_OSQP_BENCH_FORCE_DIST"):     # (world_size-1 nccl rehearsal: tests/test_gpu_multi.py)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    import osqp_solver_amd as M
    from osqp_solver_amd import problems as PR
    from osqp_solver_amd.sharding import SolutionGatherer, shard_range

    dev = torch.device("cuda", local_rank)

    def run_mode(mode, steps, warmup):
        """One measurement: returns a dict of this rank's numbers (collective calls inside)."""
        if mode == "weak":
            b0, b1 = rank * args.batch, (rank + 1) * args.batch           # every rank owns --batch QPs with its own values
        else:
            b0, b1 = shard_range(args.batch, rank, world)                 # the one batch of the metric, split
        B = b1 - b0
        pr = PR.random_box_qp(B, value_seed=1000 + b0)                    # QP b of the whole job has value seed 1000 + b
        t0 = time.time()
        solver = M.BatchSolver(pr["P"], pr["Px"], pr["q"], pr["A"], pr["Ax"], pr["l"], pr["u"], device=local_rank)
        setup_s = time.time() - t0
        st = solver.stats()
        d_l = torch.tensor(pr["l"], device=dev); d_u = torch.tensor(pr["u"], device=dev)
        d_x = torch.empty(B, pr["n"], dtype=torch.float64, device=dev)
        d_status = torch.empty(B, dtype=torch.int32, device=dev); d_iters = torch.empty_like(d_status)
        gatherer = SolutionGatherer(B, pr["n"], dev) if dist is not None else None      # static counts: exchanged once, here

        def step():
            solver.reset()
            solver.update_bounds_device(d_l, d_u)
            solver.solve_device(d_x, d_status, d_iters)
            return gatherer.gather(d_x) if gatherer is not None else d_x

        def fence():
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()

        for _ in range(warmup):
            step()
        solver.kernel_time(); solver.refactor_time()          # clear the HIP-event accumulators
        fence()
        t0 = time.perf_counter()
        dev_s = ref_s = cmp_s = 0.0
        for _ in range(steps):
            xg = step()
            ls = solver.last_solve_stats()
            dev_s += ls["device_s"]; ref_s += ls["refactor_s"]; cmp_s += ls["compact_s"]
        fence()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
        avg_ms, launches = solver.kernel_time()
        peak = solver.refactor_peak()
        f_ms, d_ms, r_launches, r_qps = solver.refactor_time()
        # the same step through the HOST-pointer boundary (BASELINE.md section 2 / SURVEY 8(d): "incl. H2D bounds and D2H
        # solutions"): bounds from host memory, solutions back to host memory.  Reported beside `value`, never as it.
        pcie = None
        if rank == 0 and mode == args.scaling:
            ksteps = max(2, min(steps, 5))
            solver.reset(); solver.update_bounds(pr["l"], pr["u"]); solver.solve(); solver.primal()      # (pinned staging allocated)
            torch.cuda.synchronize()
            tp = time.perf_counter()
            for _ in range(ksteps):
                solver.reset()
                solver.update_bounds(pr["l"], pr["u"])
                solver.solve()
                xh = solver.primal()
            tp = time.perf_counter() - tp
            pcie = {"value": B * ksteps / tp, "unit": "QPs/s", "ms_per_step": 1e3 * tp / ksteps, "steps": ksteps,
                    "bytes_per_step": 8 * B * (2 * st["m"] + st["n"]),
                    "same_solutions": bool(np.array_equal(xh, d_x.cpu().numpy()))}
            solver.kernel_time(); solver.refactor_time()
        iters = d_iters.cpu().numpy().astype(np.int64)
        status = d_status.cpu().numpy()
        tot = torch.tensor([float(B), float(iters.sum()), float(np.all(status == 1))], device=dev, dtype=torch.float64)
        if dist is not None:
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        tot = tot.cpu().numpy()
        gathered_ok = bool(xg.shape[0] == int(tot[0])) and bool(torch.equal(xg[b0 if mode == "strong" else rank * args.batch:][:B], d_x)) \
            if dist is not None else True
        return dict(B=B, b0=b0, pr=pr, solver=solver, st=st, setup_s=setup_s, elapsed=elapsed, steps=steps,
                    total_qps=int(tot[0]), total_iters=float(tot[1]), all_solved=bool(tot[2] == world),
                    avg_ms=avg_ms, launches=launches, f_ms=f_ms, d_ms=d_ms, r_launches=r_launches, r_qps=r_qps, peak=peak,
                    iters=iters, status=status, d_x=d_x, ls=solver.last_solve_stats(), dev_s=dev_s, ref_s=ref_s, cmp_s=cmp_s,
                    gathered_ok=gathered_ok, pcie=pcie)

    r = run_mode(args.scaling, args.steps, args.warmup)
    other = None
    if world > 1:                                  # the other scaling mode, measured in the same run (shorter)
        om = "strong" if args.scaling == "weak" else "weak"
        ro = run_mode(om, max(2, min(args.steps, 5)), 1)
        other = {"scaling": om, "value": ro["total_qps"] * ro["steps"] / ro["elapsed"], "unit": "QPs/s",
                 "ms_per_step": 1e3 * ro["elapsed"] / ro["steps"], "qps_per_gpu": ro["B"], "total_qps_per_step": ro["total_qps"],
                 "all_solved": ro["all_solved"]}
        ro["solver"].close()
    st, B, iters, steps = r["st"], r["B"], r["iters"], r["steps"]
    value = r["total_qps"] * steps / r["elapsed"]
    out = {
        "metric": "QPs/sec on a 1024-QP batch of box-QPs (n=512, m=1024), OSQP defaults; ADMM iters/sec alongside",
        "value": value, "unit": "QPs/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * r["elapsed"] / steps, "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "BASELINE config 3: batch of 1024 random box-QPs n=512 m=1024 (weak: per GPU; strong: in total, "
                               "block-partitioned over the GPUs), shared pattern, A=[I;G] 8 nnz/row, strictly convex banded P; "
                               "eps_abs=eps_rel=1e-3, adaptive rho interval 100",
                   "qps_per_gpu": B, "total_qps_per_step": r["total_qps"], "tile": st["tile"], "n_tiles": st["n_tiles"],
                   "nnz_L": st["nnz_L"], "nnz_L_before_tail": st["nnz_L_before_tail"],
                   "fwd_levels": st["fwd_levels"], "bwd_levels": st["bwd_levels"], "dense_tail_rows": st["dense_tail_rows"]},
        "admm_iters_per_sec": r["total_iters"] * steps / r["elapsed"],
        "iters_mean": float(iters.mean()), "iters_max": int(iters.max()),
        "all_solved": r["all_solved"],
        "setup_seconds": r["setup_s"],
        "step_breakdown_ms": {"device_iterate": 1e3 * r["dev_s"] / steps, "device_refactor": 1e3 * r["ref_s"] / steps,
                              "compaction": 1e3 * r["cmp_s"] / steps,
                              "refactors_per_step": r["ls"]["refactors"], "launches_per_step": r["ls"]["launches"]},
    }
    if r.get("pcie"):
        out["value_pcie_inclusive"] = r["pcie"]["value"]
        out["pcie_inclusive"] = dict(r["pcie"], note="the same step with bounds taken from host memory (mi_osqp_batch_update_bounds) and solutions "
                                                      "returned to host memory (mi_osqp_batch_get_primal): the unit BASELINE.md section 2 states; "
                                                      "`value` keeps inputs and outputs resident in HBM")
    if dist is not None:
        out["gather"] = {"collective": "all_gather_into_tensor (RCCL), static counts", "round_trip_ok": r["gathered_ok"]}
    if other is not None:
        out[other["scaling"]] = other
    if rank == 0:
        launches_per_step = max(1, r["launches"] // max(1, steps))
        kernel_s_per_step = r["avg_ms"] * 1e-3 * launches_per_step
        qp_iters_per_step = float(iters.sum())
        per_min, per_8d = minimal_bytes(st), algorithmic_bytes_8d(st)
        streamed = 8 * (st["fwd_slots"] + st["bwd_slots"] + st["dense_tail_slots"]) + 8 * st["N"] + 8 * (5 * st["n"] + 9 * st["m"])
        ach = per_min * qp_iters_per_step / kernel_s_per_step / 1e9 if kernel_s_per_step > 0 else 0.0
        ach8d = per_8d * qp_iters_per_step / kernel_s_per_step / 1e9 if kernel_s_per_step > 0 else 0.0
        tr = load_traffic()
        trk = tr.get("kernels", {})
        it_traffic = tr.get("bytes_per_launch") if B == HEADLINE_B else None      # the committed profile is of the 1024-QP run
        out["roofline"] = {
            "bound": "hbm", "kernel": "iterate_kernel<%d,%d>" % (st["tile"], st["threads_per_block"]), "achieved": ach, "peak": 8000.0, "unit": "GB/s",
            "frac": ach / 8000.0, "traffic": it_traffic,
            "algorithmic_bytes_per_launch": per_min * qp_iters_per_step / launches_per_step,
            "bytes_per_qp_iteration": per_min, "avg_launch_ms": r["avg_ms"], "launches_per_step": launches_per_step,
            "frac_survey_8d": ach8d / 8000.0, "bytes_per_qp_iteration_survey_8d": per_8d,
            "streamed_bytes_per_qp_iteration": streamed, "streamed_frac": ach / 8000.0 * streamed / per_min,
            "traffic_frac": (it_traffic / (r["avg_ms"] * 1e-3) / 8e12) if (it_traffic and r["avg_ms"] > 0) else None,
            "note": "achieved/frac = minimal bytes of the implemented algorithm (L before the dense tail twice, S^-1 once, vectors; "
                    "unpadded) / HIP-event launch time; frac_survey_8d = the SURVEY 8(d) formula (all of L twice), which the dense "
                    "tail undercuts; streamed_* = the padded streams the kernel reads; traffic = PMC bytes per launch of the "
                    "committed profile (profiles/hbm_traffic.json)",
        }
        out["roofline"].update(measure_stream_bandwidth(torch, dev))
        rb = refactor_bytes(st)
        kern = []
        pk_qps, pk_f, pk_t = r["peak"]                     # the largest refactorisation of the timed steps (605 QPs at the headline batch)

        def pmc_of(names):                                 # PMC bytes of the MEDIAN launch (= that refactorisation) of the listed kernels
            tot = 0.0
            for nme in names:
                cands = [v for kk, v in trk.items() if nme in kk]
                if not cands:
                    return None
                v = max(cands, key=lambda e: e.get("fetch_bytes_corrected_median_launch", 0.0))
                tot += v.get("fetch_bytes_corrected_median_launch", 0.0) + v.get("write_bytes_median_launch", 0.0)
            return tot
        for name, ms, kernels in (("factor_kernel", pk_f, ("factor_kernel",)),
                                  ("tail_kernels", pk_t, ("tail_assemble_kernel", "tail_kernel<"))):
            if not pk_qps or (name == "tail_kernels" and not st["dense_tail_rows"]):
                continue
            alg = rb[name] * pk_qps
            pmc = pmc_of(kernels) if B == HEADLINE_B else None
            kern.append({"kernel": name if name == "factor_kernel" else "tail_assemble_kernel + tail_kernel", "qps_per_launch": pk_qps,
                         "algorithmic_bytes_per_launch": alg, "launch_ms": ms,
                         "achieved": alg / (ms * 1e-3) / 1e9 if ms > 0 else 0.0, "unit": "GB/s",
                         "frac": alg / (ms * 1e-3) / 8e12 if ms > 0 else 0.0,
                         "traffic": pmc, "traffic_over_algorithmic": (pmc / alg) if pmc else None,
                         "all_launches_ms_per_step": (r["f_ms"] if name == "factor_kernel" else r["d_ms"]) / steps})
        out["roofline"]["kernels"] = kern
        if not args.no_cpu_baseline and world == 1:
            from oracle import oracle as O
            pr = r["pr"]
            cores = M.host_cores()             # min(affinity, cgroup quota): the box shows 256 CPUs, grants ~16
            sample = int(min(B, max(32, 64 * cores)))      # (16 cores: the whole batch, ~20 s of CPU work in ~1.3 s of wall time)
            rc = O.batch_solve(pr["P"], pr["Px"][:sample], pr["q"][:sample], pr["A"], pr["Ax"][:sample],
                               pr["l"][:sample], pr["u"][:sample], threads=cores, native=True)
            same = bool(np.array_equal(rc["iters"], iters[:sample]))
            err = float(np.max(np.abs(rc["x"] - r["d_x"][:sample].cpu().numpy())))
            r1 = O.batch_solve(pr["P"], pr["Px"][:8], pr["q"][:8], pr["A"], pr["Ax"][:8], pr["l"][:8], pr["u"][:8],
                               threads=1, native=True)
            out["cpu_baseline"] = {"value": sample / rc["solve_s"], "unit": "QPs/s", "cores": cores, "kind": "port",
                                   "sample": f"first {sample} QPs of the same batch, oracle solve phase on {cores} threads "
                                             f"(setup {rc['setup_s']:.2f}s excluded, as for the GPU); single-thread rate "
                                             f"{8 / r1['solve_s']:.1f} QPs/s on 8 QPs",
                                   "single_thread_qps": 8 / r1["solve_s"], "setup_seconds_sample": rc["setup_s"],
                                   "gpu_vs_cpu_max_abs_x_diff": err, "same_iteration_counts": same}
        r["solver"].close()
        if not args.no_secondary and world == 1:
            try:
                out["secondary"] = secondary(M, PR, torch, not args.no_cpu_baseline)
            except Exception as e:                          # the headline line must survive a secondary failure
                out["secondary"] = [{"error": repr(e)}]
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def measure_stream_bandwidth(torch, dev):
    """What this box's HBM delivers to plain streaming kernels, measured in the same run (report it beside the
    8 TB/s nominal): a device-to-device copy (read + write) and a triad a = b + s c (two reads + one write) over 1 GiB
    operands, best of 5, timed with events."""
    nel = 1 << 27                                   # 1 GiB of doubles per operand
    a = torch.empty(nel, dtype=torch.float64, device=dev); b = torch.ones_like(a); c = torch.ones_like(a)
    res = {}
    for name, fn, nbytes in (("measured_copy_GBps", lambda: a.copy_(b), 2 * 8 * nel),
                             ("measured_triad_GBps", lambda: torch.add(b, c, alpha=0.5, out=a), 3 * 8 * nel)):
        best = 1e30
        for _ in range(6):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e-3)
        res[name] = nbytes / best / 1e9
    del a, b, c
    torch.cuda.empty_cache()
    return res


def obstacle_scene_entry(with_cpu):
    """The reference's real workload - SQP loops that re-linearise per trajectory ([REF] src/gomp-solver.h:70-88) - through the
    continuous driver (ContinuousGOMPSolver on the per-QP entry points): 256 trajectories of a 3-link arm that must pass a
    bar, 100 waypoints.  Runs the C++ bench program built by __graft_entry__.build() as a child process (it initialises its
    own HIP context); the sequential driver on the oracle (one thread) is timed beside it on a sample."""
    exe = os.path.join(ROOT, "osqp-solver_amd", "gomp_parity_test")
    if not os.path.exists(exe):
        return {"config": "GOMP obstacle scene, continuous driver", "error": "osqp-solver_amd/gomp_parity_test not built"}
    r = subprocess.run([exe, "contbench", "256", "100", "8" if with_cpu else "0"], capture_output=True, text=True, timeout=300)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("CONTBENCH device_assembly")]
    if r.returncode != 0 or not line:
        return {"config": "GOMP obstacle scene, continuous driver", "error": (r.stdout + r.stderr)[-500:]}
    tok = line[0].split()
    kv = {tok[i]: float(tok[i + 1]) for i in range(1, len(tok) - 1, 2)}
    e = {"config": "GOMP obstacle scene (3-link arm, one collision ball, a bar to pass above, a floor): 256 trajectories x 100 waypoints, "
                   "10 horizons, SQP re-linearisation per trajectory; continuous driver (per-QP entry points, one stage per horizon)",
         "value": kv["trajectories_per_s"], "unit": "trajectories/s", "ms": 1e3 * kv["run_s"], "first_run_ms": 1e3 * kv["first_run_s"],
         "qp_solves": int(kv["qp_solves"]), "qp_updates": int(kv["qp_updates"]), "advances": int(kv["advances"]),
         "all_optimal": int(kv["optimal"]) == int(kv["trajectories"])}
    # the same run with the SQP step on the device (mi_gomp_scene: FK / Jacobians, row assembly, acceptance test, update from
    # device-resident rows): trajectories equal to round-off (device sin / cos), same solve / update counts
    try:
        r2 = subprocess.run([exe, "contbench", "256", "100", "0"], capture_output=True, text=True, timeout=300, env=dict(os.environ, GOMP_DEVICE_ASSEMBLY="1"))
        l2 = [ln for ln in r2.stdout.splitlines() if ln.startswith("CONTBENCH device_assembly")]
        if r2.returncode == 0 and l2:
            t2 = l2[0].split()
            k2 = {t2[i]: float(t2[i + 1]) for i in range(1, len(t2) - 1, 2)}
            e["device_assembly"] = {"value": k2["trajectories_per_s"], "unit": "trajectories/s", "ms": 1e3 * k2["run_s"], "qp_solves": int(k2["qp_solves"]),
                                    "qp_updates": int(k2["qp_updates"]), "all_optimal": int(k2["optimal"]) == int(k2["trajectories"])}
    except Exception as ex:          # (secondary figure: never fails the bench line)
        e["device_assembly"] = {"error": repr(ex)[:200]}
    if with_cpu and kv.get("oracle_sample", 0) > 0:
        e["cpu_baseline"] = {"value": kv["oracle_trajectories_per_s"], "unit": "trajectories/s", "cores": 1, "kind": "port",
                             "sample": "the first %d trajectories, sequential GOMPSolver on the oracle, one thread; same exit codes and "
                                       "solve / update counts, max |dx| %.1e" % (int(kv["oracle_sample"]), kv["max_dx"])}
    return e


def ur5e_scene_entry(with_cpu):
    """The reference's own robot and scene ([REF] examples/solver-example.cpp:31-70: UR5e, collision balls at wrist 3 and at the
    flange, the wall y >= -0.4) plus a bar to pass above, 128 start / goal pairs around the example's, 100 waypoints, through the
    continuous driver - with the SQP step on the host threads and on the device (mi_gomp_scene)."""
    exe = os.path.join(ROOT, "osqp-solver_amd", "gomp_parity_test")
    cfg = "GOMP UR5e scene (6 DOF, two collision balls, wall, bar): 128 trajectories x 100 waypoints, 10 horizons, continuous driver"
    if not os.path.exists(exe):
        return {"config": cfg, "error": "osqp-solver_amd/gomp_parity_test not built"}
    out = {}
    for dev in ("0", "1"):
        try:
            r = subprocess.run([exe, "contbench_ur5e", "128", "100", "4" if (with_cpu and dev == "0") else "0"], capture_output=True, text=True, timeout=300,
                               env=dict(os.environ, GOMP_DEVICE_ASSEMBLY=dev))
        except Exception as ex:
            return {"config": cfg, "error": repr(ex)[:200]}
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("CONTBENCH_UR5E device_assembly")]
        if r.returncode != 0 or not line:
            return {"config": cfg, "error": (r.stdout + r.stderr)[-500:]}
        tok = line[0].split()
        out[dev] = {tok[i]: float(tok[i + 1]) for i in range(1, len(tok) - 1, 2)}
    kv = out["0"]
    e = {"config": cfg, "value": kv["trajectories_per_s"], "unit": "trajectories/s", "ms": 1e3 * kv["run_s"], "first_run_ms": 1e3 * kv["first_run_s"],
         "qp_solves": int(kv["qp_solves"]), "qp_updates": int(kv["qp_updates"]), "all_optimal": int(kv["optimal"]) == int(kv["trajectories"]),
         "device_assembly": {"value": out["1"]["trajectories_per_s"], "unit": "trajectories/s", "ms": 1e3 * out["1"]["run_s"],
                             "qp_solves": int(out["1"]["qp_solves"]), "qp_updates": int(out["1"]["qp_updates"])}}
    if with_cpu and kv.get("oracle_sample", 0) > 0:
        e["cpu_baseline"] = {"value": kv["oracle_trajectories_per_s"], "unit": "trajectories/s", "cores": 1, "kind": "port",
                             "sample": "the first %d trajectories, sequential GOMPSolver on the oracle, one thread; same exit codes and "
                                       "solve / update counts, max |dx| %.1e" % (int(kv["oracle_sample"]), kv["max_dx"])}
    return e


def secondary(M, PR, torch, with_cpu):
    """BASELINE configs 2, 4, 5 (outside the headline timing; each a few seconds)."""
    import numpy as np
    res = []
    cores = M.host_cores()
    O = None
    if with_cpu:
        from oracle import oracle as O
    # ---- config 2: single GOMP 6-DOF trajectory QP, 50 waypoints; config 4: 256 x 7-DOF x 100 waypoints
    for name, Bq, D, W in (("config 2: single GOMP 6-DOF trajectory QP, 50 waypoints", 1, 6, 50),
                           ("config 4: batch of 256 GOMP 7-DOF trajectories, 100 waypoints", 256, 7, 100),
                           ("reference example size (examples/solver-example.cpp): single GOMP 6-DOF trajectory QP, 802 waypoints", 1, 6, 802)):
        pr = PR.gomp_batch(Bq, D, W)
        t = time.time()
        s = M.BatchSolver(pr["P"], pr["Px"], None, pr["A"], pr["Ax"], pr["l"], pr["u"])
        setup_s = time.time() - t
        st = s.stats()
        s.warm_start_x(pr["warm"]); s.solve()
        ts = []
        for _ in range(5):
            s.reset(); s.warm_start_x(pr["warm"]); torch.cuda.synchronize()
            t = time.perf_counter(); info = s.solve(); ts.append(time.perf_counter() - t)
        its = np.array([i.iter for i in info])
        e = {"config": name, "value": Bq / min(ts), "unit": "QPs/s", "ms": 1e3 * min(ts), "batch": Bq,
             "iters_mean": float(its.mean()), "iters_max": int(its.max()), "ms_per_iteration": 1e3 * min(ts) / max(1, int(its.max())),
             "all_optimal": bool(all(i.exit_code == 0 for i in info)), "setup_seconds": setup_s,
             "N": st["N"], "nnz_L": st["nnz_L"], "phases": [st["fwd_levels"], st["bwd_levels"]], "tile": st["tile"]}
        if O is not None:
            nb = min(Bq, 4 * cores)
            if Bq == 1:                       # like for like: the same warm start, one thread
                Pm, Am = PR.qp_matrices(pr, 0)
                o = O.OracleQPSolver(Pm, None, Am, pr["l"][0], pr["u"][0]); o.set_warm_start(pr["warm"][0])
                t = time.perf_counter(); o.solve(); t2 = time.perf_counter() - t
                e["cpu_baseline"] = {"value": 1.0 / t2, "unit": "QPs/s", "cores": 1, "kind": "port", "ms": 1e3 * t2, "iterations": int(o.info().iter),
                                     "sample": "the same QP, same warm start, oracle solve phase on one thread"}
                res.append(e); s.close()
                continue
            # like for like: every QP gets the warm start the GPU got; setup outside the timed region, the solves on a
            # pool of `cores` threads (the oracle's C functions run without the interpreter lock)
            from concurrent.futures import ThreadPoolExecutor
            objs = []
            for b in range(nb):
                Pm, Am = PR.qp_matrices(pr, b)
                o = O.OracleQPSolver(Pm, None, Am, pr["l"][b], pr["u"][b]); o.set_warm_start(pr["warm"][b])
                objs.append(o)
            nthreads = min(cores, nb)
            with ThreadPoolExecutor(nthreads) as pool:
                t = time.perf_counter(); sts = list(pool.map(lambda o: o.solve()[0], objs)); t2 = time.perf_counter() - t
            oit = np.array([o.info().iter for o in objs])
            e["cpu_baseline"] = {"value": nb / t2, "unit": "QPs/s", "cores": nthreads, "kind": "port", "iters_mean": float(oit.mean()),
                                 "same_iteration_counts": bool(np.array_equal(oit, its[:nb])),
                                 "sample": f"the first {nb} QPs with the same warm starts as the GPU, oracle solve phase on {nthreads} threads"}
            del objs
        res.append(e)
        s.close()
    # ---- config 5: single large sparse QP (structured: 2-D grid, see problems.grid_qp); literal size n = 99 856, m = 298 936
    # (the oracle's exact minimum degree is quadratic in the fill: it is timed beside the GPU at the reduced size only)
    for g in sorted({150, int(os.environ.get("MI_OSQP_BENCH_GRID", "316"))}):
        res.append(_grid_entry(M, PR, torch, O, g))
    res.append(obstacle_scene_entry(with_cpu))
    res.append(ur5e_scene_entry(with_cpu))
    return res


def _grid_entry(M, PR, torch, O, g):
    pr = PR.grid_qp(g)
    t = time.time()
    s = M.BatchSolver(pr["P"], pr["Px"], pr["q"], pr["A"], pr["Ax"], pr["l"], pr["u"], max_iter=200)
    setup_s = time.time() - t
    st = s.stats()
    torch.cuda.synchronize()
    t = time.perf_counter(); info = s.solve(); t1 = time.perf_counter() - t
    e = {"config": f"config 5{'' if g == 316 else ' at reduced size'}: single large sparse QP ({g} x {g} grid), n={st['n']} m={st['m']}", "value": info[0].iter / t1,
         "unit": "ADMM iterations/s", "ms": 1e3 * t1, "iterations_timed": int(info[0].iter), "ms_per_iteration": 1e3 * t1 / max(1, info[0].iter),
         "setup_seconds": setup_s, "N": st["N"], "nnz_L": st["nnz_L"], "phases": [st["fwd_levels"], st["bwd_levels"]]}
    perm = s.ordering()
    s.close()
    if g == 316:
        # roofline of the dataflow iterate kernel at the literal size: SURVEY 8(d) bytes per iteration against the launch time
        # (25 iterations per launch) and the PMC traffic of the committed profile (profiles/config5_traffic.json,
        # scripts/profile_config5_pmc.sh: FETCH_SIZE x 1.84 + WRITE_SIZE, median launch)
        alg = 2 * 8 * st["nnz_L"] + 40 * st["N"] + 8 * (5 * st["n"] + 9 * st["m"])
        e["roofline"] = {"bound": "hbm", "kernel": "iterate_kernel<1,512,true,true> (dataflow form, %d x %d threads)" % (st["solve_groups"], st["solve_group_threads"]),
                         "algorithmic_bytes_per_iteration": alg, "achieved": alg / (1e-3 * e["ms_per_iteration"]) / 1e9, "peak": 8000.0, "unit": "GB/s",
                         "frac": alg / (1e-3 * e["ms_per_iteration"]) / 8e12}
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "config5_traffic.json")))["kernels"]
            k = [v for kk, v in tr.items() if "iterate_kernel" in kk][0]
            per_launch = 1.84 * k["fetch_raw_bytes_median_launch"] + k["write_bytes_median_launch"]
            e["roofline"].update({"traffic": per_launch, "traffic_per_iteration": per_launch / 25.0, "traffic_over_algorithmic": per_launch / 25.0 / alg,
                                  "profiled_launch_ms": k["avg_ns"] * 1e-6})
        except Exception:
            e["roofline"]["traffic"] = None
    if O is not None:
        # (the oracle's own exact minimum degree is quadratic in the fill - minutes at the literal size: it factors with the
        #  elimination order the product chose; the ordering changes round-off only)
        P, A = PR.qp_matrices(pr, 0)
        o = O.OracleQPSolver(P, pr["q"][0], A, pr["l"][0], pr["u"][0], kkt_perm=perm, max_iter=200)
        t = time.perf_counter(); o.solve(); t2 = time.perf_counter() - t
        e["cpu_baseline"] = {"value": o.info().iter / t2, "unit": "ADMM iterations/s", "cores": 1, "kind": "port", "ms_per_iteration": 1e3 * t2 / max(1, o.info().iter),
                             "same_iteration_count": bool(o.info().iter == info[0].iter),
                             "sample": "same QP, 200 iterations, oracle solve phase on one thread (factor in the product's elimination order)"}
    return e


if __name__ == "__main__":
    main()
