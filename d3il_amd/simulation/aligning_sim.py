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

from ..agents import as_batched
from ..distributed import reduce_sim_counts, shard_range, world_info
from ..envs.aligning import RobotPushVecEnv, contexts_from_reference, load_test_contexts
from .base_sim import BaseSim
from .metrics import pushing_metrics

log = logging.getLogger(__name__)
N_MODES = 2


class Aligning_Sim(BaseSim):
    def __init__(self, seed: int, device: str, render: bool, n_cores: int = 1, n_contexts: int = 30, n_trajectories_per_context: int = 1,
                 if_vision: bool = False, max_steps_per_episode: int = 400, contexts: np.ndarray | None = None):
        super().__init__(seed, device, render, n_cores, if_vision)
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
        agent = as_batched(agent, n)
        agent.reset()
        quat = torch.tensor([0.0, 1.0, 0.0, 0.0], dtype=torch.float64, device=dev).expand(n, 4)
        finished = torch.zeros(n, dtype=torch.bool, device=dev)
        mode = torch.full((n,), -1, dtype=torch.int64, device=dev)
        success = torch.zeros(n, dtype=torch.bool, device=dev)
        mean_distance = torch.zeros(n, dtype=torch.float64, device=dev)
        env, flags = None, torch.zeros(0, dtype=torch.int32, device=dev)
        if n > 0:      # a rank whose shard is empty only takes part in the reductions below
            env = RobotPushVecEnv(n, device=dev, render=False, max_steps_per_episode=self.max_steps_per_episode)
            env.start()
            obs = env.reset(random=False, context=self.contexts[ctx_of.cpu().numpy()])
            des = env.robot_state().clone()                                  # pred_action = env.robot_state(), aligning_sim.py:96
            for t in range(self.max_steps_per_episode):
                obs20 = torch.cat((des, obs.to(torch.float64)), dim=1)       # np.concatenate((pred_action[:3], obs)), aligning_sim.py:99
                des_new = self._predict(agent, obs20) + obs20[:, :3]          # aligning_sim.py:101-102
                des = torch.where(finished.unsqueeze(1), des, des_new)
                action = torch.cat((des, quat), dim=1).contiguous()
                obs, _, done, info = env.step(action)
                newly = ~finished & done.bool()
                mode = torch.where(newly, info["mode"].to(torch.int64), mode)
                success = torch.where(newly, info["success"].bool(), success)
                mean_distance = torch.where(newly, info["mean_distance"], mean_distance)
                finished |= done.bool()
                if t % 16 == 15 and bool(finished.all()):                  # the only host synchronisation of the loop
                    break
            flags = env.flags[:n].clone()
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
        if env is not None:
            env.close()
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
