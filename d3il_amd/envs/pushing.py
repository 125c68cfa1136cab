"""Batched counterpart of ``Block_Push_Env``
(environments/d3il/envs/gym_pushing_env/gym_pushing/envs/pushing.py:166-500) over libd3il_rollout.

Protocol of the reference env - ``start()``, ``reset(random=False, context=...)``, ``step(action)`` returning
``(obs, reward, done, info)`` with ``info = {'mode', 'success', 'mean_distance'}``, ``robot_state()`` - for ``n_envs``
environments at once, all tensors device resident (zero-copy views of the library's HBM buffers).

Contexts.  The reference resets with ``context = [red_pos(x, y, deg), red_quat, green_pos, green_quat]`` taken from
``environments/dataset/data/pushing/test_contexts.pkl`` (pushing_sim.py:63); ``BlockContextManager.set_context`` writes
``[x, y, 0.0]`` and the quaternion into each cube's free-joint qpos (pushing.py:99-113).  Here a context is the resulting
f64[14] row ``(x, y, 0, qw, qx, qy, qz) x 2``; ``contexts_from_reference`` converts the reference's list format, and
``sample_contexts`` draws from the same spaces as ``BlockContextManager.sample`` (pushing.py:53-58,87-97).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import capi
from .avoiding import ObstacleAvoidanceVecEnv


def contexts_from_reference(ctx_list) -> np.ndarray:
    """[[red_pos(3), red_quat(4), green_pos(3), green_quat(4)], ...] (test_contexts.pkl) -> f64 [n, 14]."""
    out = np.zeros((len(ctx_list), 14))
    for i, c in enumerate(ctx_list):
        out[i, 0:2] = np.asarray(c[0], dtype=np.float64)[:2]
        out[i, 3:7] = np.asarray(c[1], dtype=np.float64)
        out[i, 7:9] = np.asarray(c[2], dtype=np.float64)[:2]
        out[i, 10:14] = np.asarray(c[3], dtype=np.float64)
    return out


def _yaw_quat(deg):
    """euler2quat([0, 0, yaw]) of utils/geometric_transformation.py:73-89 for a pure yaw: (cos(y/2), 0, 0, sin(y/2))."""
    half = np.deg2rad(deg) / 2
    q = np.zeros((len(deg), 4))
    q[:, 0], q[:, 3] = np.cos(half), np.sin(half)
    return q


def sample_contexts(n: int, seed: int = 0) -> np.ndarray:
    """Contexts drawn like BlockContextManager.sample (pushing.py:87-97): red cube x in [0.4, 0.5], green cube x in
    [0.55, 0.65], y in [-0.15, 0], yaw in [-90, 90] degrees (gym Box spaces, float32)."""
    rng = np.random.default_rng(seed)
    red = rng.uniform([0.4, -0.15, -90], [0.5, 0.0, 90], size=(n, 3)).astype(np.float32).astype(np.float64)
    green = rng.uniform([0.55, -0.15, -90], [0.65, 0.0, 90], size=(n, 3)).astype(np.float32).astype(np.float64)
    out = np.zeros((n, 14))
    out[:, 0:2], out[:, 3:7] = red[:, :2], _yaw_quat(red[:, 2])
    out[:, 7:9], out[:, 10:14] = green[:, :2], _yaw_quat(green[:, 2])
    return out


class BlockPushVecEnv(ObstacleAvoidanceVecEnv):
    task = "pushing"
    action_dim = 7
    obs_dim = 8
    default_max_steps = 400          # pushing.py:175

    def __init__(self, n_envs, device=0, render=False, n_substeps: int = 35, max_steps_per_episode: int | None = None):
        super().__init__(n_envs, device=device, render=render, n_substeps=n_substeps, max_steps_per_episode=max_steps_per_episode)
        self.mean_distance = self.info_f64[0, :self.n_envs]
        self.reward = self.info_f64[1, :self.n_envs]
        self._contexts = None

    def reset(self, mask: torch.Tensor | None = None, random: bool = False, context=None):
        """env.reset(random=False, context=...): ``context`` is f64[n_envs, 14] (numpy or tensor; see module docstring).
        With ``random=True`` contexts are sampled like BlockContextManager.sample.  A mask resets a subset."""
        if context is None:
            if not random and self._contexts is None:
                raise ValueError("Block_Push_Env.reset needs a context (or random=True)")
            if random:
                context = sample_contexts(self.n_envs, seed=int(np.random.randint(0, 2 ** 31 - 1)))
            else:
                context = self._contexts
        ctx = torch.as_tensor(context, dtype=torch.float64).to(self.device).contiguous()
        if tuple(ctx.shape) != (self.n_envs, 14):
            raise ValueError("context must have shape (%d, 14)" % self.n_envs)
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
        """Returns (obs f32[n, 8], reward f64[n], done u8[n], info) with info = dict(mode int16[n] in -1..3, success u8[n],
        mean_distance f64[n]) - pushing.py:335-339."""
        if action.device != self.device or action.dtype != torch.float64 or tuple(action.shape) != (self.n_envs, 7) or not action.is_contiguous():
            raise ValueError("action must be a contiguous float64 tensor of shape (%d, 7) on %s" % (self.n_envs, self.device))
        with torch.cuda.device(self.device):
            capi.check(self.L.d3il_step(self.h, C.c_void_p(action.data_ptr()), self._stream()))
        return self.obs, self.reward, self.done, dict(mode=self.mode, success=self.success, mean_distance=self.mean_distance)

    def box_state(self):
        """(pos f64[n, 2, 3], quat f64[n, 2, 4]) of the two cubes (scene.get_obj_pos / get_obj_quat, MjScene.py:225-247)."""
        s = self.state[capi.PUSH_STATE_BOX:capi.PUSH_STATE_BOX + 26, :self.n_envs].t().reshape(self.n_envs, 2, 13)
        return s[:, :, 0:3], s[:, :, 3:7]

    def mode_encoding(self):
        return self.mode

    def count_metrics(self, out=None):
        raise capi.D3ilError("count_metrics is Avoiding only; see simulation/pushing_sim.py for the Pushing metrics")
