"""CPU, world_size=2 over gloo: the multi-GPU layer (shard by QP index, no
data-path collective, one gather of solutions).  The per-rank "solver" here is
the oracle -- the point is the sharding/gather logic, which is backend-agnostic."""
import os
import subprocess
import sys
import textwrap

import numpy as np

from osqp_solver_amd.sharding import shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions():
    for total in (1, 7, 8, 1024, 1025):
        for world in (1, 2, 3, 8):
            r = [shard_range(total, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == total
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np, torch, torch.distributed as dist
    from osqp_solver_amd import problems as PR
    from osqp_solver_amd.sharding import shard_problem, gather_solutions, SolutionGatherer
    from oracle import oracle as O
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    prob = PR.random_box_qp(5, n=24, mg=16, nnz_per_row=3)      # 5 QPs over 2 ranks: ragged 3+2
    mine, (b0, b1) = shard_problem(prob, rank, world)
    r = O.batch_solve(mine["P"], mine["Px"], mine["q"], mine["A"], mine["Ax"], mine["l"], mine["u"])
    x, st = gather_solutions(torch.tensor(r["x"]), torch.tensor(r["status"]))
    full = O.batch_solve(prob["P"], prob["Px"], prob["q"], prob["A"], prob["Ax"], prob["l"], prob["u"])
    assert x.shape == (5, 24) and np.array_equal(x.numpy(), full["x"]), "gathered solutions differ"
    assert np.array_equal(st.numpy(), full["status"])
    g = SolutionGatherer(b1 - b0, 24, torch.device("cpu"))        # static counts (what bench.py's timed step uses)
    for _ in range(2):
        assert np.array_equal(g.gather(torch.tensor(r["x"])).numpy(), full["x"]), "static-count gather differs"
    dist.barrier()
    if rank == 0: print("GLOO_OK", b0, b1)
    dist.destroy_process_group()
""") % ROOT


def test_two_rank_gather_over_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, OMP_NUM_THREADS="1")
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    assert "GLOO_OK" in res.stdout


def test_bench_refuses_to_report_fewer_gpus_than_requested():
    """`python bench.py --gpus N` starts its own N ranks; with fewer GPUs visible it must fail loudly instead of
    printing a smaller job's number as the N-GPU point."""
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("a multi-GPU machine")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode != 0 and "GPU(s) are visible" in res.stderr
    assert not [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
