"""Replica farm over torch.distributed (backend "nccl" = RCCL over xGMI on the GPU node, "gloo" in
CPU tests).  The Brownian-dynamics path does not shard: one trajectory is a tightly coupled N-body
system that fits one GPU many times over, while the ensemble is embarrassingly parallel (the
reference runs one process per seed, 5-sim-genome/scripts/run_simulation:8-25, and averages
replicas offline, 5-sim-genome/src/contact_map/contact_map.py:14-39).  So ranks own independent
replicas; the only collectives are the broadcast of the model inputs before stepping and the gather
of a few summary statistics afterwards -- nothing inside the timed loop."""
from __future__ import annotations

import numpy as np


def _dist():
    import torch.distributed as dist
    return dist


def world():
    dist = _dist()
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _grouped():
    """True when a process group exists: the collectives then run whatever its size (a one-rank RCCL group on a one-GPU
    box executes the same broadcast / gather / all-reduce calls as the eight-rank farm); without a group they are identities."""
    dist = _dist()
    return dist.is_available() and dist.is_initialized()


def broadcast_array(arr, shape, dtype, device="cpu", src=0):
    """Broadcast a numpy array from rank `src`; other ranks pass arr=None."""
    import torch
    rank, n = world()
    if not _grouped():
        return arr
    t = torch.from_numpy(np.ascontiguousarray(arr)).to(device) if rank == src else \
        torch.empty(tuple(shape), dtype=getattr(torch, np.dtype(dtype).name), device=device)
    _dist().broadcast(t, src)
    return t.cpu().numpy()


def gather_stats(values, device="cpu", dst=0):
    """Gather a short float vector from every rank to rank `dst` -> (world, k) array (None elsewhere)."""
    import torch
    rank, n = world()
    v = torch.tensor(list(values), dtype=torch.float64, device=device)
    if not _grouped():
        return v.cpu().numpy()[None]
    out = [torch.empty_like(v) for _ in range(n)] if rank == dst else None
    _dist().gather(v, out, dst)
    return torch.stack(out).cpu().numpy() if rank == dst else None


def max_over_ranks(x, device="cpu"):
    import torch
    if not _grouped():
        return float(x)
    t = torch.tensor([float(x)], dtype=torch.float64, device=device)
    _dist().all_reduce(t, op=_dist().ReduceOp.MAX)
    return float(t.item())


def barrier():
    if _grouped():
        _dist().barrier()


def replica_seed(master_seed, rank):
    """Independent noise streams per rank (the Philox counter carries the local replica index)."""
    return int(master_seed) + 1000003 * int(rank)


def bind_to_cpu_share(local_rank, local_world, cpus_per_gpu=16):
    """Bind this rank to its GPU's share of the CPUs the process may run on (share `local_rank` of `local_world`, contiguous;
    the GPU box grants `cpus_per_gpu` = 16 CPUs per GPU): the launch threads of eight ranks -- one host thread + one stream per
    device -- then do not migrate over, or contend with, one another.  Only an affinity the local ranks SHARE is split: a set of
    at most `cpus_per_gpu` CPUs on a machine that has more is taken to be this rank's own share already (a launcher, numactl or
    a scheduler cpuset pinned it) and is kept -- cutting it again would leave 2 CPUs for the stepping thread, the packing pool and
    the writer.  `GDYN_NO_BIND=1` (bench.py --no-bind) keeps whatever the process inherited.  Returns the CPUs of the share."""
    import os
    cpus = sorted(os.sched_getaffinity(0))
    if os.environ.get("GDYN_NO_BIND") or local_world <= 1 or len(cpus) < local_world:
        return cpus
    if len(cpus) <= cpus_per_gpu and (os.cpu_count() or 0) > len(cpus):
        return cpus
    per = len(cpus) // local_world
    share = cpus[local_rank * per:(local_rank + 1) * per if local_rank < local_world - 1 else len(cpus)]
    os.sched_setaffinity(0, share)
    return share
