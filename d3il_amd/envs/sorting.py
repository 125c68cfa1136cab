"""Batched counterpart of ``Sorting_Env`` (environments/d3il/envs/gym_sorting_env/gym_sorting/envs/sorting.py:244-575,
num_boxes = 2 or 4) over libd3il_rollout.

Protocol of the reference env - ``start()``, ``reset(random=False, context=...)``, ``step(action)`` returning
``(obs, reward, done, info)`` with ``info = {'mode', 'success'}``, ``robot_state()`` - for ``n_envs`` environments at once,
all tensors device resident (zero-copy views of the library's HBM buffers).

Contexts.  ``BlockContextManager.sample`` (sorting.py:84-118) draws one (x, y, yaw) per slot from six boxes of the platform,
shuffles the six and ``set_context`` (:121-187) places the first ``num_boxes`` of them: the first half as red_1.., the second
half as blue_1.., each at ``[x, y, 0.05]`` with the yaw quaternion.  Here a context is the resulting f64 row
``(x, y, 0.05, qw, qx, qy, qz) x num_boxes`` in the order red_1.., blue_1..; ``contexts_from_reference`` converts the
reference's list format and ``sample_contexts`` draws from the same spaces.

``info['mode']`` is the reference's ``int(np.packbits(mode[:num_boxes])[0])`` (sorting.py:461-463): bit 7 - i is set while
entry i of the completion-order vector is not 0, i.e. while the i-th sorted box is not a red one (unset entries are -1).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import capi
from .avoiding import ObstacleAvoidanceVecEnv
from .pushing import _yaw_quat

_SLOTS = np.array([[0.4, -0.15, 0.5, -0.1], [0.4, -0.05, 0.5, 0.0], [0.4, 0.05, 0.5, 0.1],
                   [0.55, -0.15, 0.65, -0.1], [0.55, -0.05, 0.65, 0.0], [0.55, 0.05, 0.65, 0.1]])   # sorting.py:51-73


def contexts_from_reference(ctx_list, num_boxes: int = 4) -> np.ndarray:
    """[[[pos(x, y, deg), quat(4)] x 6], ...] (BlockContextManager.sample / *_test_contexts.pkl) -> f64 [n, 7 * num_boxes]."""
    out = np.zeros((len(ctx_list), num_boxes, 7))
    for i, c in enumerate(ctx_list):
        for k in range(num_boxes):
            out[i, k, 0:2] = np.asarray(c[k][0], dtype=np.float64)[:2]
            out[i, k, 2] = 0.05
            out[i, k, 3:7] = np.asarray(c[k][1], dtype=np.float64)
    return out.reshape(len(ctx_list), 7 * num_boxes)


def sample_contexts(n: int, num_boxes: int = 4, seed: int = 0) -> np.ndarray:
    """Contexts drawn like BlockContextManager.sample: one (x, y, yaw in [-90, 90] deg) per slot (gym Box spaces, float32),
    slots shuffled, the first ``num_boxes`` used."""
    rng = np.random.default_rng(seed)
    out = np.zeros((n, num_boxes, 7))
    for i in range(n):
        xy = rng.uniform(_SLOTS[:, :2], _SLOTS[:, 2:]).astype(np.float32).astype(np.float64)
        yaw = rng.uniform(-90, 90, size=6).astype(np.float32).astype(np.float64)
        order = rng.permutation(6)[:num_boxes]
        out[i, :, 0:2], out[i, :, 2], out[i, :, 3:7] = xy[order], 0.05, _yaw_quat(yaw[order])
    return out.reshape(n, 7 * num_boxes)


class SortingVecEnv(ObstacleAvoidanceVecEnv):
    task = "sorting"
    action_dim = 7
    num_boxes = 4
    obs_dim = 2 + 3 * num_boxes
    default_max_steps = 500          # sorting_sim.py:33

    def __init__(self, n_envs, device=0, render=False, n_substeps: int = 35, max_steps_per_episode: int | None = None, num_boxes: int = 4):
        if num_boxes not in (2, 4):
            raise NotImplementedError("this build carries the Sorting-2 and Sorting-4 scenes (the engine has one lane per cube, at most four)")
        self.num_boxes = int(num_boxes)
        self.obs_dim = 2 + 3 * self.num_boxes
        self.task = "sorting" if num_boxes == 4 else "sorting_%d" % num_boxes          # scene blob (d3il_amd/model/blobs)
        super().__init__(n_envs, device=device, render=render, n_substeps=n_substeps, max_steps_per_episode=max_steps_per_episode)
        self._contexts = None
        self.reward = torch.zeros(self.n_envs, dtype=torch.float64, device=self.device)   # get_reward is the constant 0 (sorting.py:509-511)
        self.box_row = 42
        self.warm_row = 42 + 13 * self.num_boxes
        self.task_row = self.warm_row + 6 * self.num_boxes + 9

    def reset(self, mask: torch.Tensor | None = None, random: bool = False, context=None):
        """env.reset(random=False, context=...): ``context`` is f64[n_envs, 7 * num_boxes] (numpy or tensor; see module docstring).
        With ``random=True`` contexts are sampled like BlockContextManager.sample.  A mask resets a subset."""
        if context is None:
            if not random and self._contexts is None:
                raise ValueError("Sorting_Env.reset needs a context (or random=True)")
            context = sample_contexts(self.n_envs, self.num_boxes, seed=int(np.random.randint(0, 2 ** 31 - 1))) if random else self._contexts
        ctx = torch.as_tensor(context, dtype=torch.float64).to(self.device).contiguous()
        if tuple(ctx.shape) != (self.n_envs, 7 * self.num_boxes):
            raise ValueError("context must have shape (%d, %d)" % (self.n_envs, 7 * self.num_boxes))
        mp = None
        if mask is not None:
            mask = mask.to(device=self.device, dtype=torch.uint8).contiguous()
            assert mask.numel() == self.n_envs
            mp = C.c_void_p(mask.data_ptr())
        with torch.cuda.device(self.device):
            capi.check(self.L.d3il_reset(self.h, mp, C.c_void_p(ctx.data_ptr()), self._stream()))
        self._contexts = ctx
        return self.obs

    def step(self, action: torch.Tensor):
        """Returns (obs f32[n, 14], reward f64[n] = 0, done u8[n], info) with info = dict(mode int16[n] (packbits code), success u8[n]) -
        sorting.py:444-458 (the reference's reward is the constant 0 of get_reward, :509-511)."""
        if action.device != self.device or action.dtype != torch.float64 or tuple(action.shape) != (self.n_envs, 7) or not action.is_contiguous():
            raise ValueError("action must be a contiguous float64 tensor of shape (%d, 7) on %s" % (self.n_envs, self.device))
        with torch.cuda.device(self.device):
            capi.check(self.L.d3il_step(self.h, C.c_void_p(action.data_ptr()), self._stream()))
        return self.obs, self.reward, self.done, dict(mode=self.mode, success=self.success)

    def box_state(self):
        """(pos f64[n, nb, 3], quat f64[n, nb, 4]) of the cubes, red first (scene.get_obj_pos / get_obj_quat, MjScene.py:225-247)."""
        nb = self.num_boxes
        s = self.state[self.box_row:self.box_row + 13 * nb, :self.n_envs].t().reshape(self.n_envs, nb, 13)
        return s[:, :, 0:3], s[:, :, 3:7]

    def mode_encoding(self):
        return self.mode

    def count_metrics(self, out=None):
        raise capi.D3ilError("count_metrics is Avoiding only; see simulation/metrics.py:sorting_metrics for the Sorting metrics")
