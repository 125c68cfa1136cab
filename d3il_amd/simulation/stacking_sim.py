"""Batched counterpart of ``Stacking_Sim`` (simulation/stacking_sim.py:19-257).

The reference evaluates ``n_contexts`` test contexts x ``n_trajectories_per_context`` rollouts sequentially in ``n_cores`` processes;
here every rollout is one lane of the GPU environment batch (context-major order, rollout ``c * n_trajectories + i``).  Kept from the
reference: the rollout loop (stacking_sim.py:88-109: the policy input is the LAST COMMAND - 7 desired joint positions + gripper
command, initially ``env.robot_state()`` - concatenated with the env observation; the policy output is a joint-position delta plus
the gripper command), what is recorded (``info`` of the step that returned ``done``, :118-136: the colour order string truncated to
1 / 2 / 3 letters, three success flags) and the metric tail (:143-167, :226-248: ``metrics.stacking_metrics``).

Multi-GPU: one process per GPU, contiguous shards of the rollout index range; the integer count tables are combined with one
all-reduce.
"""
from __future__ import annotations

import json
import logging
import os

import numpy as np
import torch

from ..distributed import reduce_sim_counts, shard_range, world_info
from ..envs.stacking import CubeStackingVecEnv, load_test_contexts
from ..envs.sub_batch import SubBatchSet
from ._rollout import joint_rollout
from .base_sim import BaseSim
from .metrics import stacking_metrics

log = logging.getLogger(__name__)

_DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data")
MODE_1 = {"r": 0, "g": 1, "b": 2}                                                   # stacking_sim.py:43-45
MODE_2 = {"rg": 0, "rb": 1, "gr": 2, "gb": 3, "br": 4, "bg": 5}
MODE_3 = {"rgb": 0, "rbg": 1, "grb": 2, "gbr": 3, "brg": 4, "bgr": 5}


def load_mode_prob(path: str | None = None) -> dict:
    """Prior over the six 3-box orders (environments/dataset/data/stacking/mode_prob.pkl; data copy shipped with this package)."""
    if path is None:
        with open(os.path.join(_DATA, "stacking_mode_prob.json")) as f:
            return json.load(f)
    return {k: float(v) for k, v in np.load(path, allow_pickle=True).items()}


def mode_priors(modes: dict):
    """The three prior vectors exactly as stacking_sim.py:47-62 builds them (the 2-box prior is filled with the 3-box keys' values at the
    3-box indices - kept as the reference does it)."""
    enc3, enc2 = np.zeros(6), np.zeros(6)
    for key, idx in MODE_3.items():
        enc3[idx] = modes[key]
        enc2[idx] = modes[key]
    enc1 = np.array([enc3[i] + enc3[i + 1] for i in (0, 2, 4)])
    return enc1, enc2, enc3


def _code_tables(dev):
    """mode code (n | c0 << 2 | c1 << 4 | c2 << 6) -> index into the 1- / 2- / 3-letter tables (-1: fewer letters)."""
    t1 = torch.full((256,), -1, dtype=torch.int64)
    t2, t3 = t1.clone(), t1.clone()
    for code in range(256):
        n = code & 3
        s = "".join("rgb"[(code >> (2 + 2 * i)) & 3] if ((code >> (2 + 2 * i)) & 3) < 3 else "?" for i in range(n))
        if n >= 1 and s[:1] in MODE_1:
            t1[code] = MODE_1[s[:1]]
        if n >= 2 and s[:2] in MODE_2:
            t2[code] = MODE_2[s[:2]]
        if n >= 3 and s[:3] in MODE_3:
            t3[code] = MODE_3[s[:3]]
    return t1.to(dev), t2.to(dev), t3.to(dev)


class Stacking_Sim(BaseSim):
    def __init__(self, seed: int, device: str, render: bool, n_cores: int = 1, n_contexts: int = 30, n_trajectories_per_context: int = 1,
                 max_steps_per_episode: int = 500, contexts: np.ndarray | None = None, mode_prob: dict | None = None, n_sub_batches: int = 1):
        super().__init__(seed, device, render, n_cores)
        # the reference's n_cores worker processes (stacking_sim.py:182-216) become sub-batches of the GPU batch on their own streams (envs/sub_batch.py)
        self.n_sub_batches = n_sub_batches
        self.n_contexts, self.n_trajectories_per_context = n_contexts, n_trajectories_per_context
        self.max_steps_per_episode = max_steps_per_episode
        self.test_contexts = load_test_contexts() if contexts is None else np.asarray(contexts, dtype=np.float64)
        self.modes = load_mode_prob() if mode_prob is None else dict(mode_prob)
        self.mode_encoding_1, self.mode_encoding_2, self.mode_encoding_3 = mode_priors(self.modes)
        self.last_rollout = None

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
        mean_distance = torch.zeros(n, dtype=torch.float64, device=dev)
        batches, env, flags = None, None, torch.zeros(0, dtype=torch.int32, device=dev)
        if n > 0:      # a rank whose shard is empty only takes part in the reductions below
            ctx_np = self.test_contexts[ctx_of.cpu().numpy()]

            def make_env(cnt, off):
                e = CubeStackingVecEnv(cnt, device=dev, render=False, max_steps_per_episode=self.max_steps_per_episode)
                e.start()
                e.reset(random=False, context=ctx_np[off:off + cnt])
                return e
            batches = SubBatchSet(n, self.n_sub_batches, dev, make_env)
            batches.fork_agents(agent)
            # the rollout loop of stacking_sim.py:88-109 per sub-batch (simulation/_rollout.py)
            res = joint_rollout(batches, self.max_steps_per_episode, {"mode": (torch.int64, 0), "success": (torch.bool, False), "mean_distance": (torch.float64, 0.0)})
            mode, success, mean_distance, flags = res["mode"], res["success"], res["mean_distance"], res["flags"]
            env = batches.batches[0].env
        # integer tables (stacking_sim.py:118-136, 143-151): per context, rollouts by the index of their 1- / 2- / 3-letter colour order
        t1, t2, t3 = _code_tables(dev)
        m1, m2, m3 = t1[mode & 255], t2[mode & 255], t3[mode & 255]
        s1, s2, s3 = (mode & 3) > 0, (mode & 3) > 1, success               # info['success_1'], ['success_2'], ['success']

        def table(idx, ok, n_mode):
            out = torch.zeros(self.n_contexts * n_mode, dtype=torch.int64, device=dev)
            good = ok & (idx >= 0)
            out.index_add_(0, (ctx_of * n_mode + idx.clamp_min(0)), good.to(torch.int64))
            return out

        # cal_KL counts mode_encoding[c, successes[c] == 1] == num; the mode tables are zero-initialised, so a successful rollout
        # whose order string is too short counts as index 0 (stacking_sim.py:176-183 + :149) - kept
        counts = torch.cat((table(torch.where(m1 >= 0, m1, torch.zeros_like(m1)), s1, 3), table(torch.where(m2 >= 0, m2, torch.zeros_like(m2)), s2, 6),
                            table(torch.where(m3 >= 0, m3, torch.zeros_like(m3)), s3, 6),
                            torch.stack((s1.sum(), s2.sum(), s3.sum())).to(torch.int64)))
        reduce_sim_counts(counts, env)          # the integer tables: the library's RCCL all-reduce under nccl (distributed.py)
        c = counts.cpu().numpy()
        nc = self.n_contexts
        res = stacking_metrics(c[:3 * nc].reshape(nc, 3), c[3 * nc:9 * nc].reshape(nc, 6), c[9 * nc:15 * nc].reshape(nc, 6), int(c[-3]), int(c[-2]), int(c[-1]),
                               total, self.n_trajectories_per_context, self.mode_encoding_1, self.mode_encoding_2, self.mode_encoding_3)
        self.last_rollout = dict(mode=mode, success=success, success_1=s1, success_2=s2, mean_distance=mean_distance, counts=c, shard=(lo, hi), flags=flags, metrics=res)
        log.info("Successrate %s (1 box %s, 2 boxes %s)", res["successes"], res["successes_1_box"], res["successes_2_boxes"])
        if batches is not None:
            batches.close()
        # the reference returns (successes, mode_encoding) as [n_contexts, n_trajectories] tables (stacking_sim.py:257)
        full = torch.zeros(2, total, dtype=torch.float64, device=dev)
        full[0, lo:hi] = success.to(torch.float64)
        full[1, lo:hi] = torch.where(m3 >= 0, m3, torch.zeros_like(m3)).to(torch.float64)
        if world > 1:
            import torch.distributed as dist
            dist.all_reduce(full)
        shape = (self.n_contexts, self.n_trajectories_per_context)
        return full[0].to(torch.float32).reshape(shape), full[1].to(torch.float32).reshape(shape)
