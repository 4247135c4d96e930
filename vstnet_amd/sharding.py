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


def usable_cores() -> int:
    """Cores this process may really use: the affinity mask capped by the cgroup CPU quota (a GPU box exposes every host core in
    os.cpu_count() but schedules a job on its share)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def rank_threads(world: int) -> int:
    """CPU threads one rank of `world` may use on this host: the ranks of a node share its usable cores, and N ranks x torch's
    default intra-op pool (one thread per core each) is the oversubscription that throttled the feeding thread once
    (vstnet_amd/pipeline.py)."""
    return max(1, usable_cores() // max(1, world))


VISIBLE_ENV = ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")


def visible_device_list(env=None):
    """The GPU restriction this process runs under, as the list of device tokens a child may be given, or None if there is
    none.  HIP_ / CUDA_VISIBLE_DEVICES index into what ROCR_VISIBLE_DEVICES leaves, so the first of HIP_, CUDA_, ROCR_ that is
    set names the devices as THIS process would number them (the others stay in the child's environment untouched)."""
    env = os.environ if env is None else env
    for k in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = env.get(k)
        if v is not None and v.strip() != "":
            return k, [t.strip() for t in v.split(",") if t.strip() != ""]
    return None, None


def count_gpus(env=None, kfd_root="/sys/class/kfd/kfd/topology/nodes") -> int:
    """GPUs this process may use, WITHOUT touching the HIP / HSA runtime (a parent that is about to start one child per GPU must
    not initialise a GPU itself): the visible-devices restriction if there is one, else the KFD topology nodes that have SIMDs
    (CPU nodes have simd_count 0).  0 if neither says anything (not a ROCm host)."""
    _, lst = visible_device_list(env)
    if lst is not None:
        return len(lst)
    n = 0
    try:
        for node in os.listdir(kfd_root):
            try:
                for line in open(os.path.join(kfd_root, node, "properties")):
                    if line.startswith("simd_count") and int(line.split()[1]) > 0:
                        n += 1
            except (OSError, ValueError, IndexError):
                continue
    except OSError:
        return 0
    return n


def rank_environment(rank: int, world: int, port: int | None = None, visible_device: bool = False,
                     n_devices: int | None = None) -> dict:
    """Environment of child `rank` of a self-launched node-local job: the torch.distributed.run variables (when a
    rendezvous port is given), a per-rank CPU thread cap, and — `visible_device` — a visible-devices variable so that the child
    sees exactly its own GPU as device 0 (children that need no rendezvous, e.g. video shards).  A restriction the parent
    already runs under (HIP_ / CUDA_ / ROCR_VISIBLE_DEVICES = "4,5,6,7") is honoured: child r gets the parent's r-th device,
    not physical GPU r.  n_devices (optional): the GPU count when there is no such list; with fewer GPUs than ranks (a
    rehearsal on a one-GPU box) rank r shares GPU r % n."""
    env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    n = str(rank_threads(world))
    for k in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
        env.setdefault(k, n)
    if port is not None:
        env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if visible_device:
        key, lst = visible_device_list(env)
        if lst:
            env[key] = lst[rank % len(lst)]
        else:
            env["HIP_VISIBLE_DEVICES"] = str(rank % n_devices if n_devices else rank)
        env["LOCAL_RANK"] = "0"
    return env


def launch_children(commands, environments, poll_s: float = 0.05) -> int:
    """Start one child process per (command, environment) pair and wait for all of them.  The caller has not touched a GPU
    (children do).  The first child that fails terminates the others — exactly the processes started here — and its exit
    code is returned; 0 if every child succeeded.  No re-exec, no retry."""
    import subprocess
    procs = [subprocess.Popen(list(cmd), env=env) for cmd, env in zip(commands, environments)]
    rc = 0
    pending = list(procs)
    while pending:
        for p in list(pending):
            code = p.poll()
            if code is None:
                continue
            pending.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for other in pending:
                    other.terminate()
        time.sleep(poll_s)
    return rc


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
