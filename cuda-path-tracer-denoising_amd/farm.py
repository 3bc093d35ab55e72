"""Multi-GPU: independent frames / sequences farmed across the GPUs of one node, one process and one SVGF context per
GPU, NO data-path collective (the path shards by independent sequences; within a sequence frames are serially
dependent through the history, and a spatial split would need a 2*2^L-pixel halo exchange per a-trous level —
SURVEY.md §8e).  torch.distributed (RCCL on GPU boxes, gloo in CPU tests) is used only for the barrier and for
reducing the timing: whole-job throughput = (pixels all ranks processed) / (max over ranks of the elapsed time)."""
from __future__ import annotations

import time


def shard(n_items: int, world_size: int, rank: int) -> list[int]:
    """Round-robin assignment of independent sequences to ranks; every item is owned by exactly one rank."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    return list(range(rank, n_items, world_size))


_last_local = 0.0


def last_local_seconds() -> float:
    """This rank's own elapsed time of the last timed_region() (before the MAX over ranks)."""
    return _last_local


def timed_region(step_fn, steps: int, warmup: int, sync_fn, dist=None, device=None, busy_group=None):
    """Run `warmup` untimed steps, then time exactly `steps` steps bracketed by barrier + device sync on both sides.
    Returns (max-over-ranks seconds, sum-over-ranks of the per-step work units returned by step_fn).

    busy_group (a host-side process group, gloo): the ranks first meet WITHOUT idling their GPUs — each posts an asynchronous barrier on
    that group and keeps running untimed steps until it completes — so that the bracketing barrier below finds every rank within a
    few frames of the others.  A rank that waits at a plain barrier for a slower one idles its GPU, and an MI355X that has idled for
    5 / 20 ms runs the next frames 7 / 18 % slower: the MAX over ranks would then be set by whoever arrived FIRST."""
    import torch
    for i in range(warmup):
        step_fn(i)
    if dist is not None and busy_group is not None:
        w = dist.barrier(group=busy_group, async_op=True)
        k = 0
        while not w.is_completed():
            step_fn(warmup + k)
            k += 1
            if k % 8 == 0:
                sync_fn()          # bounds the queue: at most 8 untimed frames are in flight when the last rank arrives
        w.wait()
    sync_fn()
    if dist is not None:
        dist.barrier()
    sync_fn()
    t0 = time.perf_counter()
    units = 0.0
    for i in range(steps):
        units += float(step_fn(warmup + i))
    sync_fn()
    global _last_local
    _last_local = time.perf_counter() - t0       # this rank's own steps, before it waits for the others
    if dist is not None:
        dist.barrier()
    sync_fn()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=device or "cpu")
        u = torch.tensor([units], dtype=torch.float64, device=device or "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(u, op=dist.ReduceOp.SUM)
        dt, units = float(t.item()), float(u.item())
    return dt, units
