"""Batched counterpart of ``Sorting_Sim`` (simulation/sorting_sim.py:24-213), num_box = 2 or 4.

The reference evaluates ``n_contexts`` test contexts x ``n_trajectories_per_context`` rollouts sequentially in ``n_cores``
processes; here every rollout is one group of lanes of the GPU environment batch (context-major order, rollout
``c * n_trajectories + i``).  Kept from the reference: the rollout loop (obs := desired xy || env obs, action := policy delta
+ desired xy, frozen z and quaternion, sorting_sim.py:118-130), what is recorded (``info['mode']`` / ``info['success']`` of the
step that returned ``done``, :132-133) and the metric tail (:191-213, ``metrics.sorting_metrics``).

Data.  The reference reads ``environments/dataset/data/sorting/4_test_contexts.pkl`` and ``4_mode_prob.pkl`` (:44-47); neither
file is part of its source tree.  ``contexts`` / ``mode_prob`` take them when available (``load_reference_data``); the
defaults are contexts drawn like ``BlockContextManager.sample`` (seeded) and a uniform prior over the six completion orders of
two red and two blue boxes.

Multi-GPU: one process per GPU, contiguous shards of the rollout index range; the integer tables are combined with one
all-reduce.
"""
from __future__ import annotations

import logging

import numpy as np
import torch

from ..distributed import reduce_sim_counts, shard_range, world_info
from ..envs.sorting import SortingVecEnv, contexts_from_reference, sample_contexts
from ..envs.sub_batch import SubBatchSet
from ._rollout import xy_rollout
from .base_sim import BaseSim
from .metrics import sorting_metrics

log = logging.getLogger(__name__)


def completion_order_codes(num_box: int = 4):
    """``int(np.packbits(mode)[0])`` of every order in which num_box / 2 red (0) and num_box / 2 blue (1) boxes can be completed."""
    from itertools import combinations
    codes = []
    for blue in combinations(range(num_box), num_box // 2):
        codes.append(sum(1 << (7 - i) for i in blue))
    return sorted(codes)


def load_reference_data(contexts_pkl: str, mode_prob_pkl: str, num_box: int = 4):
    """(contexts f64 [n, 7 num_box], {mode code: prior probability}) from the reference's two data files."""
    ctx = contexts_from_reference(np.load(contexts_pkl, allow_pickle=True), num_box)
    modes = np.load(mode_prob_pkl, allow_pickle=True)
    return ctx, {int(k): float(v) for k, v in modes.items()}


class Sorting_Sim(BaseSim):
    def __init__(self, seed: int, device: str, render: bool, n_cores: int = 1, n_contexts: int = 30, n_trajectories_per_context: int = 1,
                 num_box: int = 4, if_vision: bool = False, max_steps_per_episode: int = 500, contexts: np.ndarray | None = None,
                 mode_prob: dict | None = None, n_sub_batches: int = 1):
        super().__init__(seed, device, render, n_cores, if_vision)
        # the reference's n_cores worker processes (sorting_sim.py:160-189) become sub-batches of the GPU batch on their own streams (envs/sub_batch.py)
        self.n_sub_batches = n_sub_batches
        if num_box not in (2, 4):
            raise NotImplementedError("this build carries the Sorting-2 and Sorting-4 scenes")
        self.n_contexts, self.n_trajectories_per_context = n_contexts, n_trajectories_per_context
        self.max_steps_per_episode, self.num_box = max_steps_per_episode, num_box
        self.test_contexts = sample_contexts(max(n_contexts, 60), num_box, seed=seed) if contexts is None else np.asarray(contexts, dtype=np.float64)
        if mode_prob is None:
            codes = completion_order_codes(num_box)
            mode_prob = {c: 1.0 / len(codes) for c in codes}
        self.modes = dict(mode_prob)
        self.mode_keys = np.array(list(self.modes.keys()))           # sorting_sim.py:49-57
        self.n_mode = len(self.mode_keys)
        self.mode_encoding = torch.tensor([self.modes[k] for k in self.mode_keys])
        self.last_rollout = None

    def _predict(self, agent, obs_in: torch.Tensor) -> torch.Tensor:
        return agent.predict_batch(obs_in).to(device=obs_in.device, dtype=torch.float64).reshape(obs_in.shape[0], 2)

    def test_agent(self, agent):
        log.info("Starting trained model evaluation")
        rank, world = world_info()
        total = self.n_contexts * self.n_trajectories_per_context
        lo, hi = shard_range(total, rank, world)
        n = hi - lo
        dev = torch.device(self.device)
        ctx_of = torch.arange(lo, hi, device=dev) // self.n_trajectories_per_context
        mode = torch.zeros(n, dtype=torch.int64, device=dev)
        success = torch.zeros(n, dtype=torch.bool, device=dev)
        env, batches, flags = None, None, torch.zeros(0, dtype=torch.int32, device=dev)
        if n > 0:      # a rank whose shard is empty (fewer rollouts than ranks) only takes part in the reductions below
            ctx_np = self.test_contexts[ctx_of.cpu().numpy()]

            def make_env(cnt, off):
                e = SortingVecEnv(cnt, device=dev, render=False, max_steps_per_episode=self.max_steps_per_episode, num_boxes=self.num_box)
                e.start()
                e.reset(random=False, context=ctx_np[off:off + cnt])
                return e
            batches = SubBatchSet(n, self.n_sub_batches, dev, make_env)
            batches.fork_agents(agent)
            # the rollout loop of sorting_sim.py:118-133 per sub-batch (simulation/_rollout.py)
            res = xy_rollout(batches, self.max_steps_per_episode, {"mode": (torch.int64, 0), "success": (torch.bool, False)}, predict=self._predict)
            mode, success, flags = res["mode"], res["success"], res["flags"]
            env = batches.batches[0].env
        # integer table: per context, successful rollouts whose mode code is the k-th key of the prior; number of successes
        keys = torch.as_tensor(self.mode_keys, dtype=torch.int64, device=dev)
        hit = (mode.unsqueeze(1) == keys.unsqueeze(0)) & success.unsqueeze(1)                 # [n, n_mode]
        counts = torch.zeros(self.n_contexts * self.n_mode + 1, dtype=torch.int64, device=dev)
        slot = (ctx_of.unsqueeze(1) * self.n_mode + torch.arange(self.n_mode, device=dev).unsqueeze(0)).reshape(-1)
        counts[:-1].index_add_(0, slot, hit.reshape(-1).to(torch.int64))
        counts[-1] = success.sum()
        mode_hist = torch.bincount(mode.clamp(0, 255), minlength=256)       # all rollouts, by final mode code (diagnostics)
        reduce_sim_counts(counts, env)          # the integer tables: the library's RCCL all-reduce under nccl (distributed.py)
        reduce_sim_counts(mode_hist, env)
        c = counts.cpu().numpy()
        success_rate, entropy, kl, score = sorting_metrics(c[:-1].reshape(self.n_contexts, self.n_mode), int(c[-1]), total, self.n_trajectories_per_context,
                                                           self.mode_encoding.numpy())
        self.last_rollout = dict(mode=mode, success=success, counts=c, mode_hist=mode_hist.cpu().numpy(), shard=(lo, hi), flags=flags)
        log.info("Successrate %s entropy %s KL %s", success_rate, entropy, kl)
        if batches is not None:
            batches.close()
        # the quantities the reference logs (sorting_sim.py:209-212)
        return {"score": score, "Metrics/successes": success_rate, "Metrics/KL": kl, "Metrics/entropy": entropy}
