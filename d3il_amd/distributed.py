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
    host channel that hands rank 0's 128-byte unique id to the other ranks (any backend).

    Set-up is collective and every step of it is AGREED on before the next one starts, so a rank that cannot resolve RCCL (or whose
    ncclGetUniqueId / ncclCommInitRank fails) makes ALL ranks raise the same error instead of leaving the others blocked in a broadcast or
    inside ncclCommInitRank."""

    def __init__(self, device: torch.device | int):
        import ctypes as C
        from . import capi
        self.L, self.C = capi.load(), C
        rank, world = world_info()
        self.rank, self.world = rank, world
        self.device = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        self.comm = None
        # 1. does every rank have an RCCL?  (no communicator is made yet)
        if not self._all_ok(bool(self.L.d3il_rccl_available())):
            raise RuntimeError("LibraryComm: RCCL could not be resolved on every rank")
        # 2. the unique id: rank 0 broadcasts either the 128 bytes or an error marker (never raises before the broadcast)
        uid = (C.c_char * 128)()
        box = [None]
        if rank == 0:
            rc = self.L.d3il_comm_unique_id(C.byref(uid))
            box = [bytes(uid) if rc == 0 else "error: %s" % self.L.d3il_last_error().decode()]
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        if not isinstance(box[0], (bytes, bytearray)):
            raise RuntimeError("LibraryComm: rank 0 could not make an RCCL unique id (%s)" % (box[0],))
        uid = (C.c_char * 128).from_buffer_copy(box[0])
        # 3. ncclCommInitRank on every rank, then agreement on its outcome
        comm = C.c_void_p()
        with torch.cuda.device(self.device):
            rc = self.L.d3il_comm_init(C.byref(uid), rank, world, self.device.index or 0, C.byref(comm))
        err = None if rc == 0 else self.L.d3il_last_error().decode()
        if rc == 0:
            self.comm = comm
        if not self._all_ok(rc == 0):
            self.close()
            raise RuntimeError("LibraryComm: ncclCommInitRank failed on some rank%s" % ("" if err is None else " (here: %s)" % err))

    def _all_ok(self, ok: bool) -> bool:
        if self.world <= 1:
            return ok
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(int(t.item()))

    def ranks(self) -> int:
        """ncclCommCount: the number of ranks RCCL itself sees in this communicator."""
        from . import capi
        n = self.C.c_int(0)
        capi.check(self.L.d3il_comm_count(self.comm, self.C.byref(n)))
        return int(n.value)

    def reduce(self, handle, table: torch.Tensor) -> torch.Tensor:
        """In-place sum of an int64 device tensor over all ranks; ``handle`` may be None (a rank without environments)."""
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


_AUTO_COMM: dict = {}


def auto_comm(device) -> "LibraryComm | None":
    """The process-wide library communicator the Sim classes reduce their count tables with: made once, when torch.distributed runs on
    the "nccl" (= RCCL) backend with more than one rank; None otherwise (single process, or the gloo test mode, where the tables go through
    torch.distributed).  ``D3IL_LIB_REDUCE=0`` forces the torch.distributed path.  Collective on first use: every rank must call it."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() <= 1:
        return None
    if dist.get_backend() != "nccl" or os.environ.get("D3IL_LIB_REDUCE", "1") != "1":
        return None
    dev = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
    key = (dev.index or 0, dist.get_world_size())
    if key not in _AUTO_COMM:
        try:
            _AUTO_COMM[key] = LibraryComm(dev)
        except RuntimeError as exc:      # agreed on by all ranks (see LibraryComm): everybody falls back together
            import warnings
            warnings.warn("library RCCL communicator unavailable (%s); count tables are reduced through torch.distributed" % exc)
            _AUTO_COMM[key] = None
    return _AUTO_COMM[key]


LAST_REDUCTION = "none (single process)"


def reduce_counts(counts: torch.Tensor, comm: "LibraryComm | None" = None, handle=None) -> torch.Tensor:
    """Sum int64 count tensors over all ranks (in place).  With a LibraryComm the all-reduce is the library's own RCCL call
    (d3il_reduce_metrics; ``handle`` = the env's handle or None); otherwise torch.distributed's (backend "nccl" = RCCL, "gloo" in the CPU
    tests); no-op for a single process without a communicator.  ``LAST_REDUCTION`` names the path the last call took."""
    global LAST_REDUCTION
    assert counts.dtype == torch.int64
    if comm is not None:
        LAST_REDUCTION = "libd3il_rollout d3il_reduce_metrics: one RCCL ncclAllReduce(sum, int64), %d ranks" % comm.ranks()
        return comm.reduce(handle, counts)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        LAST_REDUCTION = "torch.distributed all_reduce (%s)" % dist.get_backend()
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    else:
        LAST_REDUCTION = "none (single process)"
    return counts


def reduce_sim_counts(counts: torch.Tensor, env) -> torch.Tensor:
    """What the four Sim classes call at the end of a rollout (the ONE exchange step): int64 tables summed by the library's RCCL
    all-reduce under the nccl backend, by torch.distributed otherwise.  ``env`` is the rank's VecEnv or None (empty shard)."""
    comm = auto_comm(counts.device) if counts.is_cuda else None
    return reduce_counts(counts, comm, None if env is None else env.h)
