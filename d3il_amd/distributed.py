"""Env sharding across GPUs and the single metric reduction.

Environments are independent for the whole rollout, so rank r owns the contiguous index range
``shard_range(n, r, world)`` and runs its own model copy / state / RNG stream (Philox counters are offset by
the global env index, so results do not depend on ``world``).  The only exchange is one all-reduce (sum) of
int64 counts at the end - backend "nccl" is RCCL on ROCm, "gloo" on CPU for tests.  Integer sums make the
metrics bit-exact and order independent.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def shard_range(n_total: int, rank: int, world: int):
    lo = (n_total * rank + world - 1) // world
    hi = (n_total * (rank + 1) + world - 1) // world
    return lo, hi


def world_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def init_from_env(backend: str | None = None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* when launched by torch.distributed.run."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 or dist.is_initialized():
        return world_info()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    dist.init_process_group(backend=backend, rank=int(os.environ["RANK"]), world_size=world)
    return world_info()


def reduce_counts(counts: torch.Tensor) -> torch.Tensor:
    """Sum int64 count tensors over all ranks (in place); no-op for a single process."""
    assert counts.dtype == torch.int64
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    return counts
