"""Multi-GPU sharding of the batch: instances are independent NLPs, so the batch is cut into contiguous blocks,
one per rank (one process per GPU); there is NO collective in the data path.  torch.distributed (RCCL on the GPU
box, gloo in the CPU tests) is used only to gather the controls on rank 0 and to reduce three statistics."""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(batch: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous block [lo, hi) of rank `rank`; sizes differ by at most one."""
    base, rem = divmod(batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_rows(local: torch.Tensor, batch: int, rank: int, world: int) -> torch.Tensor | None:
    """All ranks' row blocks -> (batch, cols) on every rank (all_gather of equally padded blocks)."""
    if world == 1:
        return local
    cols = local.shape[1]
    size = (batch + world - 1) // world
    # (gloo, the backend of the CPU tests and of the one-GPU rehearsal of bench.py, gathers host tensors only; RCCL gathers in place)
    dev = local.device
    coll = torch.device("cpu") if (local.is_cuda and dist.get_backend() == "gloo") else dev
    pad = torch.zeros(size, cols, dtype=local.dtype, device=coll)
    pad[: local.shape[0]] = local.to(coll)
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    if coll != dev:
        parts = [q.to(dev) for q in parts]
    out = []
    for r in range(world):
        lo, hi = shard_range(batch, r, world)
        out.append(parts[r][: hi - lo])
    return torch.cat(out, dim=0)


def reduce_stats(max_iters: int, n_failed: int, seconds: float, device=None):
    """(max over ranks of iterations, sum of failures, max of time): the only reduction of a tick."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return max_iters, n_failed, seconds
    t = torch.tensor([float(max_iters), float(seconds)], dtype=torch.float64, device=device)
    s = torch.tensor([float(n_failed)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    return int(t[0].item()), int(s[0].item()), float(t[1].item())
