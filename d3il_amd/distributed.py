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


class LibraryComm:
    """RCCL communicator owned by libd3il_rollout (d3il_comm_* / d3il_reduce_metrics, include/d3il_rollout.h): the metric reduction is
    issued by the library itself, on the caller's HIP stream, as ONE ncclAllReduce(sum, int64) over xGMI.  torch.distributed is only the
    host channel that hands rank 0's 128-byte unique id to the other ranks (any backend)."""

    def __init__(self, device: torch.device | int):
        import ctypes as C
        from . import capi
        self.L, self.C = capi.load(), C
        rank, world = world_info()
        self.rank, self.world = rank, world
        self.device = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        uid = (C.c_char * 128)()
        if rank == 0:
            capi.check(self.L.d3il_comm_unique_id(C.byref(uid)))
        box = [bytes(uid)]
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        uid = (C.c_char * 128).from_buffer_copy(box[0])
        self.comm = C.c_void_p()
        with torch.cuda.device(self.device):
            capi.check(self.L.d3il_comm_init(C.byref(uid), rank, world, self.device.index or 0, C.byref(self.comm)))

    def reduce(self, handle, table: torch.Tensor) -> torch.Tensor:
        from . import capi
        assert table.dtype == torch.int64 and table.is_cuda and table.is_contiguous()
        with torch.cuda.device(self.device):
            capi.check(self.L.d3il_reduce_metrics(handle, self.comm, self.C.c_void_p(table.data_ptr()), table.numel(),
                                                  self.C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
        return table

    def close(self):
        if self.comm:
            self.L.d3il_comm_destroy(self.comm)
            self.comm = None


def reduce_counts(counts: torch.Tensor, comm: "LibraryComm | None" = None, handle=None) -> torch.Tensor:
    """Sum int64 count tensors over all ranks (in place).  With a LibraryComm (and the env's handle) the all-reduce is the library's own RCCL
    call (d3il_reduce_metrics); otherwise torch.distributed's (backend "nccl" = RCCL, "gloo" in the CPU tests); no-op for a single process
    without a communicator."""
    assert counts.dtype == torch.int64
    if comm is not None:
        return comm.reduce(handle, counts)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    return counts
