"""Batched counterpart of ``Aligning_Sim`` (simulation/aligning_sim.py:30-204), state observations.

Every rollout (context c, trajectory i) is one lane of the GPU environment batch, context-major.  Kept from the reference: the rollout loop
(obs := desired xyz || env obs, action := policy delta + desired xyz - the policy commands x, y AND z -, frozen quaternion [0, 1, 0, 0],
aligning_sim.py:98-104), what is recorded (``info`` of the step that returned ``done``, :106-108) and the metric tail with two behaviour modes
(:176-204, ``metrics.pushing_metrics(n_modes=2)``).

Multi-GPU: one process per GPU, contiguous shards; the integer tables go through ``distributed.reduce_sim_counts`` (the library's RCCL
all-reduce under nccl), the f64 distance sum through torch.distributed.
"""
from __future__ import annotations

import logging

import numpy as np
import torch

from ..distributed import reduce_sim_counts, shard_range, world_info
from ..envs.aligning import RobotPushVecEnv, contexts_from_reference, load_test_contexts
from ..envs.sub_batch import SubBatchSet
from ._rollout import xyz_rollout
from .base_sim import BaseSim
from .metrics import pushing_metrics

log = logging.getLogger(__name__)
N_MODES = 2


class Aligning_Sim(BaseSim):
    def __init__(self, seed: int, device: str, render: bool, n_cores: int = 1, n_contexts: int = 30, n_trajectories_per_context: int = 1,
                 if_vision: bool = False, max_steps_per_episode: int = 400, contexts: np.ndarray | None = None, n_sub_batches: int = 1):
        super().__init__(seed, device, render, n_cores, if_vision)
        # the reference's n_cores worker processes (aligning_sim.py:125-160) become sub-batches of the GPU batch on their own streams (envs/sub_batch.py)
        self.n_sub_batches = n_sub_batches
        if if_vision:
            raise NotImplementedError("the batched rollout path serves state observations (SURVEY section 2: vision is out of scope)")
        self.n_contexts = n_contexts
        self.n_trajectories_per_context = n_trajectories_per_context
        self.max_steps_per_episode = max_steps_per_episode
        self.contexts = load_test_contexts() if contexts is None else np.asarray(contexts, dtype=np.float64)
        self.last_rollout = None

    def load_reference_contexts(self, path: str):
        """The reference's own pickle (environments/dataset/data/aligning/test_contexts.pkl, aligning_sim.py:18-22)."""
        self.contexts = contexts_from_reference(np.load(path, allow_pickle=True))

    def _predict(self, agent, obs20: torch.Tensor) -> torch.Tensor:
        return agent.predict_batch(obs20).to(device=obs20.device, dtype=torch.float64).reshape(obs20.shape[0], 3)

    def test_agent(self, agent):
        log.info("Starting trained model evaluation")
        rank, world = world_info()
        total = self.n_contexts * self.n_trajectories_per_context
        lo, hi = shard_range(total, rank, world)
        n = hi - lo
        dev = torch.device(self.device)
        ctx_of = torch.arange(lo, hi, device=dev) // self.n_trajectories_per_context
        mode = torch.full((n,), -1, dtype=torch.int64, device=dev)
        success = torch.zeros(n, dtype=torch.bool, device=dev)
        mean_distance = torch.zeros(n, dtype=torch.float64, device=dev)
        batches, env, flags = None, None, torch.zeros(0, dtype=torch.int32, device=dev)
        if n > 0:      # a rank whose shard is empty only takes part in the reductions below
            ctx_np = self.contexts[ctx_of.cpu().numpy()]

            def make_env(cnt, off):
                e = RobotPushVecEnv(cnt, device=dev, render=False, max_steps_per_episode=self.max_steps_per_episode)
                e.start()
                e.reset(random=False, context=ctx_np[off:off + cnt])
                return e
            batches = SubBatchSet(n, self.n_sub_batches, dev, make_env)
            batches.fork_agents(agent)
            # the rollout loop of aligning_sim.py:96-108 per sub-batch (simulation/_rollout.py)
            res = xyz_rollout(batches, self.max_steps_per_episode, {"mode": (torch.int64, -1), "success": (torch.bool, False), "mean_distance": (torch.float64, 0.0)},
                              predict=self._predict)
            mode, success, mean_distance, flags = res["mode"], res["success"], res["mean_distance"], res["flags"]
            env = batches.batches[0].env
        # integer tables: mode counts of the successful rollouts per context, number of successes; f64 distance sum
        counts = torch.zeros(self.n_contexts * N_MODES + 1, dtype=torch.int64, device=dev)
        ok = success & (mode >= 0)
        counts[:-1] = torch.bincount((ctx_of * N_MODES + mode.clamp_min(0))[ok], minlength=self.n_contexts * N_MODES)
        counts[-1] = success.sum()
        dist_sum = torch.nan_to_num(mean_distance, nan=0.0).sum().reshape(1)
        reduce_sim_counts(counts, env)
        if world > 1:
            import torch.distributed as dist
            dist.all_reduce(dist_sum)
        c = counts.cpu().numpy()
        success_rate, entropy, mode_probs = pushing_metrics(c[:-1].reshape(self.n_contexts, N_MODES), int(c[-1]), total, self.n_trajectories_per_context, n_modes=N_MODES)
        self.last_rollout = dict(mode=mode, success=success, mean_distance=mean_distance, counts=c, shard=(lo, hi), success_rate=success_rate, entropy=entropy,
                                 mode_probs=mode_probs, mean_distance_all=float(dist_sum.item()) / total, flags=flags, score=0.5 * (success_rate + entropy))
        log.info("Successrate %s entropy %s mean distance %s", success_rate, entropy, float(dist_sum.item()) / total)
        if batches is not None:
            batches.close()
        # the reference returns (success_rate, mode_encoding[n_contexts, n_trajectories]) (aligning_sim.py:205); its tables are zero-initialised, so a
        # rollout that never reported a mode reads 0 there.  The full tables stay available in self.last_rollout (+ "tables" below).
        full = torch.zeros(3, total, dtype=torch.float64, device=dev)
        full[0, lo:hi], full[1, lo:hi], full[2, lo:hi] = success.to(torch.float64), mode.clamp_min(0).to(torch.float64), torch.nan_to_num(mean_distance, nan=0.0)
        if world > 1:
            import torch.distributed as dist
            dist.all_reduce(full)
        shape = (self.n_contexts, self.n_trajectories_per_context)
        self.last_rollout["tables"] = dict(successes=full[0].to(torch.float32).reshape(shape), mode_encoding=full[1].to(torch.float32).reshape(shape),
                                           mean_distance=full[2].to(torch.float32).reshape(shape))
        return success_rate, self.last_rollout["tables"]["mode_encoding"]
