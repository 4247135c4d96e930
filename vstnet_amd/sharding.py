"""Frame sharding across the GPUs of one node.

Stylised frames are independent, so rank r of N owns a contiguous range of the frame index and nothing is
exchanged on the data path.  torch.distributed (RCCL on ROCm, gloo on CPU) is used only to line the ranks up
(barrier) and to take the slowest rank's clock.
"""
from __future__ import annotations

import os
import time


def shard_range(n_frames: int, rank: int, world: int):
    """Contiguous, balanced shard [lo, hi) of frames 0..n_frames-1 for `rank` (keeps video order per GPU:
    config 5's 300 frames on 8 GPUs -> 38,38,38,38,37,37,37,37)."""
    if not (0 <= rank < world) or n_frames < 0:
        raise ValueError("bad shard arguments")
    base, extra = divmod(n_frames, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def dist_env():
    """(rank, local_rank, world) from the torch.distributed.run environment (1-process defaults)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def timed_steps(step, steps: int, warmup: int, sync, world: int, device=None, return_all: bool = False):
    """W untimed warm-up steps, then exactly K steps bracketed by barrier + sync on both sides;
    returns the MAX over ranks of the elapsed seconds (the bench contract) — with ``return_all`` also every
    rank's own time between its two syncs (for the per-rank report)."""
    import torch
    import torch.distributed as dist
    for _ in range(warmup):
        step()
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    own = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    every = [own]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device or "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        if return_all:
            mine = torch.tensor([own], dtype=torch.float64, device=device or "cpu")
            gathered = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(gathered, mine)
            every = [float(g.item()) for g in gathered]
    return (elapsed, every) if return_all else elapsed
