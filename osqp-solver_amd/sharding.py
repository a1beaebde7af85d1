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


class SolutionGatherer:
    """The gather of solutions with STATIC shard sizes: the per-rank counts are exchanged once, here (outside any
    timed step); gather() is then one all_gather_into_tensor into a preallocated buffer and involves no host
    synchronisation.  Ragged shards (counts differ by one) are padded to the largest and trimmed."""

    def __init__(self, count, n, device, dtype=None, group=None):
        import torch
        import torch.distributed as dist
        self.group, self.n = group, n
        self.world = dist.get_world_size(group)
        dtype = dtype or torch.float64
        cnt = torch.tensor([count], device=device, dtype=torch.int64)
        counts = torch.zeros(self.world, device=device, dtype=torch.int64)
        dist.all_gather_into_tensor(counts, cnt, group=group)
        self.counts = [int(c) for c in counts.cpu()]
        self.count, self.mx = count, max(self.counts)
        self.even = all(c == self.mx for c in self.counts)
        self.buf = torch.empty((self.world * self.mx, n), device=device, dtype=dtype)
        self.pad = None if count == self.mx else torch.zeros((self.mx, n), device=device, dtype=dtype)

    def gather(self, x_local):
        import torch
        import torch.distributed as dist
        src = x_local
        if self.pad is not None:
            self.pad[:self.count].copy_(x_local); src = self.pad
        dist.all_gather_into_tensor(self.buf, src.contiguous(), group=self.group)
        if self.even:
            return self.buf
        return torch.cat([self.buf[r * self.mx: r * self.mx + c] for r, c in enumerate(self.counts)])


def shard_problem(prob, rank, world):
    """Slice a problems.py batch dict down to this rank's QPs."""
    B = prob["Ax"].shape[0]
    b0, b1 = shard_range(B, rank, world)
    out = dict(prob)
    for k in ("Px", "Ax", "q", "l", "u", "warm"):
        if k in prob and prob[k] is not None:
            out[k] = np.ascontiguousarray(prob[k][b0:b1])
    return out, (b0, b1)
