"""Multi-GPU layer: the batch of independent QPs is the shard axis (one process
per GPU, torch.distributed; backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in CPU tests).  The ADMM iterate needs no exchange at all; the only
collective is the gather of solutions BASELINE.json's north_star names."""
import numpy as np


def shard_range(total, rank, world):
    """Contiguous block partition of `total` QPs: the first (total % world) ranks
    get one extra.  Returns (begin, end)."""
    base, rem = divmod(total, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def gather_solutions(x_local, status_local=None, group=None):
    """all_gather of per-rank solution blocks [B_r, n] (B_r may differ by one
    between ranks; blocks are padded to the max and trimmed).  Returns the
    [B_total, n] tensor on every rank (and the gathered status vector)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    cnt = torch.tensor([x_local.shape[0]], device=x_local.device, dtype=torch.int64)
    counts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(counts, cnt, group=group)
    counts = [int(c.item()) for c in counts]
    mx = max(counts)
    pad = x_local
    if x_local.shape[0] < mx:
        pad = torch.cat([x_local, x_local.new_zeros((mx - x_local.shape[0],) + tuple(x_local.shape[1:]))])
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad.contiguous(), group=group)
    x = torch.cat([b[:c] for b, c in zip(bufs, counts)])
    if status_local is None:
        return x
    sp = status_local
    if sp.shape[0] < mx:
        sp = torch.cat([sp, sp.new_zeros(mx - sp.shape[0])])
    sb = [torch.empty_like(sp) for _ in range(world)]
    dist.all_gather(sb, sp.contiguous(), group=group)
    return x, torch.cat([b[:c] for b, c in zip(sb, counts)])


def shard_problem(prob, rank, world):
    """Slice a problems.py batch dict down to this rank's QPs."""
    B = prob["Ax"].shape[0]
    b0, b1 = shard_range(B, rank, world)
    out = dict(prob)
    for k in ("Px", "Ax", "q", "l", "u", "warm"):
        if k in prob and prob[k] is not None:
            out[k] = np.ascontiguousarray(prob[k][b0:b1])
    return out, (b0, b1)
