"""Batched counterpart of ``Pushing_Sim`` (simulation/pushing_sim.py:27-178).

The reference evaluates ``n_contexts`` test contexts x ``n_trajectories_per_context`` rollouts sequentially in
``n_cores`` processes; here every rollout is one lane of the GPU environment batch (context-major order, rollout
``c * n_trajectories + i``).  Kept from the reference: the rollout loop (obs := desired xy || env obs, action := policy
delta + desired xy, frozen z and quaternion, pushing_sim.py:61-79), what is recorded (``info`` of the step that returned
``done``, :81-83) and the metric tail (:140-167, ``metrics.pushing_metrics``).

Multi-GPU: one process per GPU, contiguous shards of the rollout index range; the integer tables (mode counts per
context, success count) and the f64 distance sum are combined with one all-reduce each.
"""
from __future__ import annotations

import logging
import os

import numpy as np
import torch

from ..distributed import reduce_sim_counts, shard_range, world_info
from ..envs.pushing import BlockPushVecEnv, contexts_from_reference
from ..envs.sub_batch import SubBatchSet
from ._rollout import xy_rollout
from .base_sim import BaseSim
from .metrics import pushing_metrics

log = logging.getLogger(__name__)

_DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data", "pushing_test_contexts.npy")


def load_test_contexts(path: str | None = None) -> np.ndarray:
    """The reference's evaluation contexts as f64 [60, 14].  ``path`` may point at the reference's own
    ``environments/dataset/data/pushing/test_contexts.pkl``; by default the copy shipped as data with this package is used."""
    if path is None:
        return np.load(_DATA)
    return contexts_from_reference(np.load(path, allow_pickle=True))


class Pushing_Sim(BaseSim):
    def __init__(self, seed: int, device: str, render: bool, n_cores: int = 1, n_contexts: int = 30,
                 n_trajectories_per_context: int = 1, max_steps_per_episode: int = 400, contexts: np.ndarray | None = None, n_sub_batches: int = 1):
        super().__init__(seed, device, render, n_cores)
        # the reference's n_cores worker processes (pushing_sim.py:129-165) become sub-batches of the GPU batch on their own streams (envs/sub_batch.py)
        self.n_sub_batches = n_sub_batches
        self.n_contexts = n_contexts
        self.n_trajectories_per_context = n_trajectories_per_context
        self.max_steps_per_episode = max_steps_per_episode
        self.contexts = load_test_contexts() if contexts is None else np.asarray(contexts, dtype=np.float64)
        self.last_rollout = None

    def _predict(self, agent, obs10: torch.Tensor) -> torch.Tensor:
        return agent.predict_batch(obs10).to(device=obs10.device, dtype=torch.float64).reshape(obs10.shape[0], 2)

    def test_agent(self, agent):
        log.info("Starting trained model evaluation")
        rank, world = world_info()
        total = self.n_contexts * self.n_trajectories_per_context
        lo, hi = shard_range(total, rank, world)
        n = hi - lo
        dev = torch.device(self.device)
        ctx_of = torch.arange(lo, hi, device=dev) // self.n_trajectories_per_context          # context index of each rollout
        mode = torch.full((n,), -1, dtype=torch.int64, device=dev)
        success = torch.zeros(n, dtype=torch.bool, device=dev)
        mean_distance = torch.zeros(n, dtype=torch.float64, device=dev)
        env, batches, flags = None, None, torch.zeros(0, dtype=torch.int32, device=dev)
        if n > 0:      # a rank whose shard is empty (fewer rollouts than ranks) only takes part in the reductions below
            ctx_np = self.contexts[ctx_of.cpu().numpy()]

            def make_env(cnt, off):
                e = BlockPushVecEnv(cnt, device=dev, render=False, max_steps_per_episode=self.max_steps_per_episode)
                e.start()
                e.reset(random=False, context=ctx_np[off:off + cnt])
                return e
            batches = SubBatchSet(n, self.n_sub_batches, dev, make_env)
            batches.fork_agents(agent)
            # the rollout loop of pushing_sim.py:69-84 per sub-batch (simulation/_rollout.py)
            res = xy_rollout(batches, self.max_steps_per_episode, {"mode": (torch.int64, -1), "success": (torch.bool, False), "mean_distance": (torch.float64, 0.0)}, predict=self._predict)
            mode, success, mean_distance, flags = res["mode"], res["success"], res["mean_distance"], res["flags"]
            env = batches.batches[0].env
        # integer tables: mode counts of the successful rollouts per context, number of successes; f64 distance sum
        counts = torch.zeros(self.n_contexts * 4 + 1, dtype=torch.int64, device=dev)
        ok = success & (mode >= 0)
        counts[:-1] = torch.bincount((ctx_of * 4 + mode.clamp_min(0))[ok], minlength=self.n_contexts * 4)
        counts[-1] = success.sum()
        dist_sum = mean_distance.sum().reshape(1)
        reduce_sim_counts(counts, env)          # the integer tables: the library's RCCL all-reduce under nccl (distributed.py)
        if world > 1:
            import torch.distributed as dist
            dist.all_reduce(dist_sum)
        c = counts.cpu().numpy()
        success_rate, entropy, mode_probs = pushing_metrics(c[:-1].reshape(self.n_contexts, 4), int(c[-1]), total, self.n_trajectories_per_context)
        self.last_rollout = dict(mode=mode, success=success, mean_distance=mean_distance, counts=c, shard=(lo, hi),
                                 success_rate=success_rate, entropy=entropy, mode_probs=mode_probs,
                                 mean_distance_all=float(dist_sum.item()) / total, flags=flags)
        log.info("Successrate %s entropy %s mean distance %s", success_rate, entropy, float(dist_sum.item()) / total)
        if batches is not None:
            batches.close()
        # the reference returns the full [n_contexts, n_trajectories] tables (pushing_sim.py:178): every rank fills its slice of a
        # zero table and the slices are summed (one more small all-reduce, outside the rollout)
        full = torch.zeros(3, total, dtype=torch.float64, device=dev)
        full[0, lo:hi], full[1, lo:hi], full[2, lo:hi] = success.to(torch.float64), mode.to(torch.float64), mean_distance
        if world > 1:
            import torch.distributed as dist
            dist.all_reduce(full)
        shape = (self.n_contexts, self.n_trajectories_per_context)
        return full[0].to(torch.float32).reshape(shape), full[1].to(torch.float32).reshape(shape), full[2].to(torch.float32).reshape(shape)
