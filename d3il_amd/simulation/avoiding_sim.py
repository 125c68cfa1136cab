"""Batched counterpart of ``Avoiding_Sim`` (simulation/avoiding_sim.py:20-144).

The reference spawns ``n_cores`` processes with one MuJoCo env each and rolls ``n_trajectories`` episodes
out sequentially; here every rollout is one lane of the GPU environment batch and all of them advance
together.  The rollout loop (obs || desired-xy concat, delta -> absolute action, frozen z and quaternion,
avoiding_sim.py:51-66) and the metric tail (:126-144) are kept; agents are called once per step on the whole
batch through ``predict_batch`` when they provide it, else row by row through the reference's
``predict(np.ndarray) -> np.ndarray[1, 2]`` protocol (agents/base_agent.py:110-122).

Multi-GPU: launched one process per GPU (torch.distributed, backend nccl = RCCL); every rank rolls out its
contiguous shard of rollouts and the integer success / mode-histogram counts are summed with ONE all-reduce
(int64, 514 elements), so the result is bit-exact and independent of the number of ranks.
"""
from __future__ import annotations

import logging

import numpy as np
import torch

from ..distributed import reduce_sim_counts, shard_range, world_info
from ..envs.avoiding import ObstacleAvoidanceVecEnv
from ..agents import as_batched
from .base_sim import BaseSim
from .metrics import avoiding_metrics

log = logging.getLogger(__name__)


class Avoiding_Sim(BaseSim):
    def __init__(self, seed: int, device: str, render: bool, n_cores: int = 1, n_trajectories: int = 30,
                 max_steps_per_episode: int = 250):
        super().__init__(seed, device, render, n_cores)
        self.n_trajectories = n_trajectories
        self.max_steps_per_episode = max_steps_per_episode
        self.last_rollout = None

    def _predict(self, agent, obs4: torch.Tensor) -> torch.Tensor:
        return agent.predict_batch(obs4).to(device=obs4.device, dtype=torch.float64).reshape(obs4.shape[0], 2)

    def test_agent(self, agent):
        log.info("Starting trained model evaluation")
        rank, world = world_info()
        lo, hi = shard_range(self.n_trajectories, rank, world)
        n = hi - lo
        dev = torch.device(self.device)
        agent = as_batched(agent, n)
        agent.reset()
        quat = torch.tensor([0.0, 1.0, 0.0, 0.0], dtype=torch.float64, device=dev).expand(n, 4)
        finished = torch.zeros(n, dtype=torch.bool, device=dev)
        mode_code = torch.zeros(n, dtype=torch.int32, device=dev)
        success = torch.zeros(n, dtype=torch.bool, device=dev)
        # the reference stores c_pos in a 150-row buffer and crashes on longer episodes (avoiding_sim.py:73,87);
        # this buffer is sized max_steps + 1 instead (documented divergence)
        c_pos = torch.zeros(n, self.max_steps_per_episode + 1, 2, dtype=torch.float64, device=dev)
        n_pos = torch.ones(n, dtype=torch.int64, device=dev)
        env = None
        if n > 0:      # a rank whose shard is empty (n_trajectories < world size) only takes part in the reductions below
            env = ObstacleAvoidanceVecEnv(n, device=dev, render=False, max_steps_per_episode=self.max_steps_per_episode)
            env.start()
            obs = env.reset()
            pred_action = env.robot_state().clone()                       # TCP xyz, avoiding_sim.py:53
            fixed_z = pred_action[:, 2:3].clone()
            des_xy = pred_action[:, :2].clone()
            c_pos[:, 0] = env.robot_state()[:, :2]
            rows = torch.arange(n, device=dev)
            for t in range(self.max_steps_per_episode):
                # avoiding_sim.py:61: np.concatenate((f64 desired xy, f32 obs)) -> f64[4]
                obs4 = torch.cat((des_xy, obs.to(torch.float64)), dim=1)
                delta = self._predict(agent, obs4)
                des_new = delta + obs4[:, :2]                              # avoiding_sim.py:64
                des_xy = torch.where(finished.unsqueeze(1), des_xy, des_new)
                action = torch.cat((des_xy, fixed_z, quat), dim=1).contiguous()
                obs, _, done, (mode, succ) = env.step(action)
                active = ~finished
                # no boolean-mask indexing (its size is a host sync): finished lanes rewrite their last row with itself
                c_pos[rows, n_pos] = torch.where(active.unsqueeze(1), env.robot_state()[:, :2], c_pos[rows, n_pos])
                n_pos += active.to(torch.int64)
                newly = active & done.bool()
                mode_code = torch.where(newly, mode.to(torch.int32), mode_code)
                success = torch.where(newly, succ.bool(), success)
                finished |= done.bool()
                if t % 16 == 15 and bool(finished.all()):                  # the only host synchronisation of the loop
                    break
        counts = torch.zeros(514, dtype=torch.int64, device=dev)
        counts[0] = n
        counts[1] = success.sum()
        counts[2:] = torch.bincount(mode_code[success].to(torch.int64), minlength=512)
        reduce_sim_counts(counts, env)
        c = counts.cpu().numpy()
        success_rate, entropy = avoiding_metrics(int(c[0]), int(c[1]), c[2:])
        self.last_rollout = dict(success=success, mode_code=mode_code, c_pos=c_pos, n_pos=n_pos, counts=c, shard=(lo, hi))
        log.info("Successrate %s entropy %s", success_rate, entropy)
        # the reference returns the success flags of all trajectories (avoiding_sim.py:144): shards are put together
        successes = torch.zeros(self.n_trajectories, dtype=torch.float32, device=dev)
        successes[lo:hi] = success.to(torch.float32)
        if world > 1:
            import torch.distributed as dist
            dist.all_reduce(successes)
        if env is not None:
            env.close()
        return successes, entropy
