"""CPU, world_size 2 over gloo: the N>1 host path of bench.py — contiguous frame shards with no data-path
collective, barrier + max-over-ranks timing."""
import os
import socket
import sys
import time

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vstnet_amd.sharding import shard_range, timed_steps

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions():
    for n, w in ((32, 8), (300, 8), (5, 8), (1, 1), (0, 4), (7, 2)):
        covered = []
        sizes = []
        for r in range(w):
            lo, hi = shard_range(n, r, w)
            assert 0 <= lo <= hi <= n
            covered += list(range(lo, hi))
            sizes.append(hi - lo)
        assert covered == list(range(n))                 # contiguous, ordered, exactly once
        assert max(sizes) - min(sizes) <= 1
    assert [shard_range(300, r, 8)[1] - shard_range(300, r, 8)[0] for r in range(8)] == [38] * 4 + [37] * 4
    with pytest.raises(ValueError):
        shard_range(4, 4, 4)


def _worker(rank, world, port, out):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(7, rank, world)
    done = []

    def step():                                           # stand-in for "stylise my shard"; rank 1 is slower
        done.append(list(range(lo, hi)))
        time.sleep(0.02 * (rank + 1))

    elapsed = timed_steps(step, steps=3, warmup=1, sync=lambda: None, world=world)
    # the only communication is the timing reduce; gather shard lists here just to check the partition
    shards = [None] * world
    dist.all_gather_object(shards, (lo, hi, len(done)))
    if rank == 0:
        torch.save({"elapsed": elapsed, "shards": shards}, out)
    dist.destroy_process_group()


def test_two_ranks_gloo(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "res.pt")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    res = torch.load(out, weights_only=False)
    assert res["shards"] == [(0, 4, 4), (4, 7, 4)]       # 1 warm-up + 3 timed steps each, disjoint shards
    assert res["elapsed"] >= 3 * 0.04 * 0.9               # MAX over ranks: rank 1 sleeps 40 ms per step
