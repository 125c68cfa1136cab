"""Sub-batches: a process's environments stepped as S independent environment batches on S HIP streams.

The reference spreads an evaluation over ``n_cores`` worker processes, each stepping its own slice of the contexts / rollouts
(simulation/avoiding_sim.py:87-124, pushing_sim.py:129-165, sorting_sim.py:160-189: ``mp.Process`` per core, shared result tables).
The counterpart on one GPU: the rollouts of a rank are cut into S contiguous sub-batches; each owns an environment handle
(its own state buffers, tally and kernel launches), a HIP stream and a clone of the agent's per-episode state.  A step launch lasts as
long as its slowest workgroup, and launches of different streams overlap - sub-batch B's physics runs while sub-batch A's policy does
(DESIGN section 18.10).  Rollouts are independent of each other, so for a policy that computes every row from that row alone (deterministic, or seeded
per rollout) the integer result tables are those of one batch (tests/test_subbatch_sims.py); a policy that draws from the process-wide torch generator
(DDPM / BESO: torch.randn per predict call) draws in another order per sub-batch - the same distribution, not the same table.

Used by the Sim classes (``n_sub_batches=``) and by ``bench.py`` (``--sub-batches``).
"""
from __future__ import annotations

import copy
import os
import warnings

import torch

MIN_ENVS = 64          # one wavefront of environments: below that a sub-batch only adds launches
DEFAULT_HW_QUEUES = 4   # what the HIP runtime maps streams onto when GPU_MAX_HW_QUEUES is not set
MAX_CONCURRENT = 4     # dispatches of different streams the MI355X runs side by side (measured: 8 / 16 / 32 sub-batches of the contact tasks take 2 / 4 / 8 rounds of
                       # four - Pushing contact regime 7.6 / 18.0 / 27.5 / 46.5 ms per step at S = 4 / 8 / 16 / 32 although all 256 workgroups would fit the chip and
                       # GPU_MAX_HW_QUEUES was raised; profiles/r05/subbatch_sweep/): more sub-batches than this only add launches
CUS = 256              # MI355X: the Avoiding kernel's third ("serve") wave pays while the GPU holds at most one workgroup per CU


def plan(n: int, n_sub_batches: int, min_envs: int = MIN_ENVS):
    """[(offset, count)] of at most ``n_sub_batches`` contiguous sub-batches of ``n`` environments, every one at least ``min_envs`` wide
    (fewer sub-batches when n is small), sizes differing by at most one."""
    s = max(1, min(int(n_sub_batches), n // min_envs if n >= min_envs else 1))
    base, extra = divmod(n, s)
    out, off = [], 0
    for i in range(s):
        c = base + (1 if i < extra else 0)
        out.append((off, c))
        off += c
    return out


def ensure_hw_queues(n_streams: int):
    """Streams only overlap when the runtime gives them hardware queues of their own: ``GPU_MAX_HW_QUEUES`` (read once, when HIP initialises)
    must be >= the number of sub-batches.  Before initialisation the variable is raised; afterwards a too small value is reported."""
    cur = os.environ.get("GPU_MAX_HW_QUEUES")
    have = int(cur) if cur and cur.isdigit() else DEFAULT_HW_QUEUES
    if have >= n_streams:
        return have
    if not torch.cuda.is_initialized():
        os.environ["GPU_MAX_HW_QUEUES"] = str(n_streams)
        return n_streams
    warnings.warn("%d sub-batches but GPU_MAX_HW_QUEUES = %d was fixed when HIP initialised: their streams share hardware queues and will not fully "
                  "overlap (export GPU_MAX_HW_QUEUES=%d before the first GPU call)" % (n_streams, have, n_streams))
    return have


def fork_agent(agent):
    """A clone of a batched agent for one sub-batch: networks, scalers and everything immutable are shared, per-episode state (history deques,
    action-chunk buffers, non-parameter tensors; agents.RowwiseAgent's rule) is copied - the sub-batches call ``predict_batch`` in turn and must not
    see each other's histories.  Agents that know better provide ``fork()``."""
    if hasattr(agent, "fork"):
        return agent.fork()
    from ..agents import _is_lane_state
    c = copy.copy(agent)
    for k, v in list(vars(agent).items()):
        if k.startswith("_parameters") or k.startswith("_modules") or k.startswith("_buffers"):
            continue                      # torch.nn.Module internals stay shared
        if _is_lane_state(v):
            setattr(c, k, copy.deepcopy(v))
    return c


class SubBatch:
    """One sub-batch: offset / count inside the rank's rollout range, its environment, stream and agent clone."""
    __slots__ = ("index", "offset", "n", "env", "stream", "own_stream", "agent", "state")

    def __init__(self, index, offset, n, env, stream, own_stream):
        self.index, self.offset, self.n, self.env, self.stream, self.own_stream = index, offset, n, env, stream, own_stream
        self.agent, self.state = None, None


class SubBatchSet:
    """Owns the S environment handles and streams of a rank.

    make_env(count, offset) -> environment (constructed under the sub-batch's stream; the set then binds the stream to the handle, so every library
    call of that environment - step, reset, auto-reset, tally - is launched on it without a stream context per call).
    """

    def __init__(self, n: int, n_sub_batches: int, device, make_env, min_envs: int = MIN_ENVS):
        self.device = torch.device(device)
        self.parts = plan(n, n_sub_batches, min_envs)
        self.n = n
        S = len(self.parts)
        if S > 1:
            ensure_hw_queues(S)
        if S > MAX_CONCURRENT:
            warnings.warn("%d sub-batches: the GPU runs %d dispatches of different streams at a time, the others wait their turn (DESIGN section 19.6)" % (S, MAX_CONCURRENT))
        self.default_stream = torch.cuda.current_stream(self.device)
        self.batches = []
        for i, (off, cnt) in enumerate(self.parts):
            own = S > 1
            stream = torch.cuda.Stream(self.device) if own else self.default_stream
            with torch.cuda.stream(stream):
                env = make_env(cnt, off)
                if own:
                    env.bind_stream(stream)
                # Avoiding: the serve wave is taken while the whole GPU holds at most one workgroup per CU - the library sees one sub-batch, the set all
                if own and hasattr(env, "set_option") and type(env).__name__ == "ObstacleAvoidanceVecEnv":
                    env.set_option("serve_wave_max_workgroups", CUS // S)
            self.batches.append(SubBatch(i, off, cnt, env, stream, own))
        if S > 1:
            for b in self.batches:      # set-up work of the sub-batch streams is ordered before whatever the caller does next on its stream
                self.default_stream.wait_stream(b.stream)

    def __len__(self):
        return len(self.batches)

    def __iter__(self):
        return iter(self.batches)

    def each(self, fn):
        """fn(sub_batch) for every sub-batch with torch's current stream set to the sub-batch's (the policy's torch kernels go there); the caller's
        stream is restored afterwards."""
        if len(self.batches) == 1:
            fn(self.batches[0])
            return
        try:
            for b in self.batches:
                torch.cuda.set_stream(b.stream)
                fn(b)
        finally:      # also when fn raises (a library error, an assert inside a policy): the process must not stay on a sub-batch stream
            torch.cuda.set_stream(self.default_stream)

    def join(self):
        """The caller's stream waits for everything queued on the sub-batch streams (no host synchronisation)."""
        if len(self.batches) > 1:
            for b in self.batches:
                self.default_stream.wait_stream(b.stream)

    def fork_agents(self, agent):
        """Per-sub-batch agent clones (one sub-batch: the agent itself).  An agent that wants to know which rollouts its rows are (per-rollout seeds,
        goals) implements ``set_rollout_range(offset, count)``: rows 0 .. count-1 of its batch are rollouts offset .. offset+count-1 of the rank."""
        from ..agents import as_batched
        for b in self.batches:
            a = agent if len(self.batches) == 1 else fork_agent(agent)
            if hasattr(a, "set_rollout_range"):
                a.set_rollout_range(b.offset, b.n)
            b.agent = as_batched(a, b.n)
            b.agent.reset()

    def close(self):
        for b in self.batches:
            if b.env is not None:
                b.env.close()
                b.env = None
