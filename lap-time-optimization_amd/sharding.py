"""Multi-GPU sharding of the batch: instances are independent NLPs, so the batch is cut into contiguous blocks,
one per rank (one process per GPU); there is NO collective in the data path.  torch.distributed (RCCL on the GPU
box, gloo in the CPU tests) is used only to gather the controls on rank 0 and to reduce three statistics."""
from __future__ import annotations

import threading

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


class TickPipeline:
    """Per-tick collective beside free-running parts (SplitMPC.run_ticks on a rank of a multi-GPU job).

    The parts of a rank tick at their own pace, each in its own host thread, but every rank has to issue the SAME sequence of
    collectives, so one thread (the caller of `consume`) issues them: the parts write the output of tick t to slot t % depth of a
    ring; when ALL parts of the rank have finished tick t, `consume` calls fn(t) (the gather of that slot), and a part starts tick
    t + depth - which writes the same slot again - only after fn(t) has returned.  The parts therefore run at most depth - 1 ticks
    ahead of the collective, and no part ever waits for another rank inside its tick.

    before_tick / after_tick are the hooks of SplitMPC.run_ticks (called in the parts' threads); `fail` wakes everybody up when one
    side has raised, so that neither waits for ever."""

    def __init__(self, n_parts: int, depth: int = 2):
        if n_parts < 1 or depth < 1:
            raise ValueError("TickPipeline: n_parts and depth must be positive")
        self.n_parts, self.depth = int(n_parts), int(depth)
        self._cond = threading.Condition()
        self._done = [0] * self.n_parts  # ticks finished per part
        self._consumed = -1              # last tick whose slot has been consumed
        self._failure: list[BaseException] = []

    def before_tick(self, part: int, t: int):
        with self._cond:
            self._cond.wait_for(lambda: self._consumed >= t - self.depth or self._failure)
            if self._failure:
                raise RuntimeError("TickPipeline: the other side failed") from self._failure[0]

    def after_tick(self, part: int, t: int):
        with self._cond:
            self._done[part] = t + 1
            self._cond.notify_all()

    def fail(self, e: BaseException):
        with self._cond:
            self._failure.append(e)
            self._cond.notify_all()

    @property
    def failure(self):
        return self._failure[0] if self._failure else None

    def consume(self, n_ticks: int, fn):
        """fn(t) for t = 0 .. n_ticks - 1, each as soon as every part has finished tick t.  Returns False if a part failed."""
        for t in range(n_ticks):
            with self._cond:
                self._cond.wait_for(lambda: min(self._done) >= t + 1 or self._failure)
                if self._failure:
                    return False
            try:
                fn(t)
            except BaseException as e:
                self.fail(e)
                raise
            with self._cond:
                self._consumed = t
                self._cond.notify_all()
        return True
