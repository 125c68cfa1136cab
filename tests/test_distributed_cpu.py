"""N > 1 path on CPU: world_size-2 gloo processes shard the rollouts and reduce integer counts once."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from d3il_amd.distributed import shard_range
from d3il_amd.simulation.metrics import avoiding_metrics


def test_shards_partition_the_env_range():
    for n in (1, 7, 64, 4096, 4099):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def _synthetic(n):
    rng = np.random.default_rng(5)
    succ = rng.uniform(size=n) < 0.6
    code = (1 << rng.integers(0, 2, n)) | (1 << (2 + rng.integers(0, 3, n))) | (1 << (5 + rng.integers(0, 4, n)))
    return succ, code


def _worker(rank, world, port, n, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from d3il_amd.distributed import init_from_env, reduce_counts, shard_range as sr
    init_from_env("gloo")
    succ, code = _synthetic(n)
    lo, hi = sr(n, rank, world)
    counts = torch.zeros(514, dtype=torch.int64)
    counts[0] = hi - lo
    counts[1] = int(succ[lo:hi].sum())
    counts[2:] = torch.as_tensor(np.bincount(code[lo:hi][succ[lo:hi]], minlength=512))
    reduce_counts(counts)
    q.put((rank, counts.numpy()))
    torch.distributed.destroy_process_group()


def test_two_rank_reduction_equals_single_process():
    n, world = 1001, 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    [p.start() for p in procs]
    res = dict(q.get(timeout=120) for _ in range(world))
    [p.join(60) for p in procs]
    succ, code = _synthetic(n)
    ref = np.zeros(514, dtype=np.int64)
    ref[0], ref[1] = n, succ.sum()
    ref[2:] = np.bincount(code[succ], minlength=512)
    for r in range(world):
        assert np.array_equal(res[r], ref)
    assert avoiding_metrics(int(ref[0]), int(ref[1]), ref[2:]) == avoiding_metrics(int(res[0][0]), int(res[0][1]), res[0][2:])
