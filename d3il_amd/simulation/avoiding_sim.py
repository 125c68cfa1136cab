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
from ..envs.sub_batch import SubBatchSet
from .base_sim import BaseSim
from .metrics import avoiding_metrics

log = logging.getLogger(__name__)


class Avoiding_Sim(BaseSim):
    def __init__(self, seed: int, device: str, render: bool, n_cores: int = 1, n_trajectories: int = 30,
                 max_steps_per_episode: int = 250, n_sub_batches: int = 1):
        super().__init__(seed, device, render, n_cores)
        # the reference's n_cores worker processes (avoiding_sim.py:87-124) become sub-batches of the GPU batch on their own streams (envs/sub_batch.py)
        self.n_sub_batches = n_sub_batches
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
        mode_code = torch.zeros(n, dtype=torch.int32, device=dev)
        success = torch.zeros(n, dtype=torch.bool, device=dev)
        # the reference stores c_pos in a 150-row buffer and crashes on longer episodes (avoiding_sim.py:73,87);
        # this buffer is sized max_steps + 1 instead (documented divergence)
        c_pos = torch.zeros(n, self.max_steps_per_episode + 1, 2, dtype=torch.float64, device=dev)
        n_pos = torch.ones(n, dtype=torch.int64, device=dev)
        env, batches = None, None
        if n > 0:      # a rank whose shard is empty (n_trajectories < world size) only takes part in the reductions below
            def make_env(cnt, off):
                e = ObstacleAvoidanceVecEnv(cnt, device=dev, render=False, max_steps_per_episode=self.max_steps_per_episode)
                e.start()
                e.reset()
                return e
            batches = SubBatchSet(n, self.n_sub_batches, dev, make_env)
            batches.fork_agents(agent)

            class _Lanes:
                pass

            def begin(b):
                st = b.state = _Lanes()
                m = b.n
                st.quat = torch.tensor([0.0, 1.0, 0.0, 0.0], dtype=torch.float64, device=dev).expand(m, 4)
                st.finished = torch.zeros(m, dtype=torch.bool, device=dev)
                st.mode_code = torch.zeros(m, dtype=torch.int32, device=dev)
                st.success = torch.zeros(m, dtype=torch.bool, device=dev)
                st.c_pos = torch.zeros(m, self.max_steps_per_episode + 1, 2, dtype=torch.float64, device=dev)
                st.n_pos = torch.ones(m, dtype=torch.int64, device=dev)
                pred_action = b.env.robot_state().clone()                       # TCP xyz, avoiding_sim.py:53
                st.fixed_z = pred_action[:, 2:3].clone()
                st.des_xy = pred_action[:, :2].clone()
                st.c_pos[:, 0] = b.env.robot_state()[:, :2]
                st.rows = torch.arange(m, device=dev)
                st.obs = b.env.obs

            def step(b):
                st, e = b.state, b.env
                # avoiding_sim.py:61: np.concatenate((f64 desired xy, f32 obs)) -> f64[4]
                obs4 = torch.cat((st.des_xy, st.obs.to(torch.float64)), dim=1)
                delta = self._predict(b.agent, obs4)
                des_new = delta + obs4[:, :2]                              # avoiding_sim.py:64
                st.des_xy = torch.where(st.finished.unsqueeze(1), st.des_xy, des_new)
                action = torch.cat((st.des_xy, st.fixed_z, st.quat), dim=1).contiguous()
                st.obs, _, done, (mode, succ) = e.step(action)
                active = ~st.finished
                # no boolean-mask indexing (its size is a host sync): finished lanes rewrite their last row with itself
                st.c_pos[st.rows, st.n_pos] = torch.where(active.unsqueeze(1), e.robot_state()[:, :2], st.c_pos[st.rows, st.n_pos])
                st.n_pos += active.to(torch.int64)
                newly = active & done.bool()
                st.mode_code = torch.where(newly, mode.to(torch.int32), st.mode_code)
                st.success = torch.where(newly, succ.bool(), st.success)
                st.finished |= done.bool()

            batches.each(begin)
            for t in range(self.max_steps_per_episode):
                batches.each(step)
                if t % 16 == 15:                                           # the only host synchronisation of the loop
                    batches.join()
                    if bool(torch.stack([b.state.finished.all() for b in batches]).all()):
                        break
            batches.join()
            torch.cuda.synchronize(dev)
            mode_code = torch.cat([b.state.mode_code for b in batches])
            success = torch.cat([b.state.success for b in batches])
            c_pos = torch.cat([b.state.c_pos for b in batches])
            n_pos = torch.cat([b.state.n_pos for b in batches])
            env = batches.batches[0].env
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
        if batches is not None:
            batches.close()
        return successes, entropy
