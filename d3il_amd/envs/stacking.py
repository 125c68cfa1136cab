"""Batched counterpart of ``CubeStacking_Env``
(environments/d3il/envs/gym_stacking_env/gym_stacking/envs/stacking.py:135-481) over libd3il_rollout.

Protocol of the reference env - ``start()``, ``reset(random=False, context=...)``, ``step(action)`` returning
``(obs, reward, done, info)`` with ``info = {'mode', 'success', 'success_1', 'success_2', 'mean_distance'}``, ``robot_state()`` - for
``n_envs`` environments at once, all tensors device resident (zero-copy views of the library's HBM buffers).

Differences from the Cartesian tasks, as in the reference: the action is JOINT space - 7 joint targets for the joint PD controller
plus the gripper command (open iff ``action[7] > 0.075``, stacking.py:331-346); 30 physics sub-steps per env step (:138); the
observation is the three boxes' ``(x, y, z, tan yaw)`` (:228-277) and ``robot_state()`` returns the 7 joint positions and the gripper
width (:199-212), which the rollout loop concatenates (stacking_sim.py:93-101).

Contexts.  ``environments/dataset/data/stacking/test_contexts.pkl`` holds 100 x [red, green, blue, target] x [pos(x, y, deg), quat];
``BlockContextManager.set_context`` writes ``[x, y, 0]`` and the quaternion of the three boxes into their free-joint qpos
(stacking.py:99-125; the target entry is not used).  Here a context is the resulting f64[21] row ``(x, y, 0, qw, qx, qy, qz) x 3``.

``info['mode']`` is the string of box colours in the order they reached the target zone (:395-419); it is returned as an integer
code ``n | c0 << 2 | c1 << 4 | c2 << 6`` (n letters, colours 0 r, 1 g, 2 b) - ``mode_string`` decodes it.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from .. import capi
from .avoiding import ObstacleAvoidanceVecEnv

_DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data")


def contexts_from_reference(ctx_list) -> np.ndarray:
    """[[pos(x, y, deg), quat] x 4 (red, green, blue, target), ...] (test_contexts.pkl) -> f64 [n, 21]."""
    out = np.zeros((len(ctx_list), 21))
    for i, c in enumerate(ctx_list):
        for b in range(3):
            out[i, 7 * b:7 * b + 2] = np.asarray(c[b][0], dtype=np.float64)[:2]
            out[i, 7 * b + 3:7 * b + 7] = np.asarray(c[b][1], dtype=np.float64)
    return out


def load_test_contexts(path: str | None = None) -> np.ndarray:
    """The reference's 100 evaluation contexts as f64 [100, 21] (data copy shipped with this package, or the reference's pickle)."""
    if path is None:
        return np.load(os.path.join(_DATA, "stacking_test_contexts.npy"))
    return contexts_from_reference(np.load(path, allow_pickle=True))


def sample_contexts(n: int, seed: int = 0) -> np.ndarray:
    """Contexts drawn like BlockContextManager.sample (stacking.py:52-97): red x in [0.35, 0.45], y in [-0.25, -0.15]; green x in
    [0.35, 0.45], y in [-0.1, 0]; blue x in [0.55, 0.6], y in [-0.2, 0]; yaw in [-90, 90] degrees (gym Box spaces, float32)."""
    rng = np.random.default_rng(seed)
    lo = np.array([[0.35, -0.25, -90], [0.35, -0.1, -90], [0.55, -0.2, -90]])
    hi = np.array([[0.45, -0.15, 90], [0.45, 0.0, 90], [0.6, 0.0, 90]])
    out = np.zeros((n, 21))
    for b in range(3):
        p = rng.uniform(lo[b], hi[b], size=(n, 3)).astype(np.float32).astype(np.float64)
        half = np.deg2rad(p[:, 2]) / 2
        out[:, 7 * b:7 * b + 2] = p[:, :2]
        out[:, 7 * b + 3], out[:, 7 * b + 6] = np.cos(half), np.sin(half)
    return out


def mode_string(code: int) -> str:
    n = code & 3
    return "".join("rgb"[(code >> (2 + 2 * i)) & 3] for i in range(n))


class CubeStackingVecEnv(ObstacleAvoidanceVecEnv):
    task = "stacking"
    action_dim = 8
    obs_dim = 12
    default_max_steps = 1000         # configs/stacking_config.yaml:84

    def __init__(self, n_envs, device=0, render=False, n_substeps: int = 30, max_steps_per_episode: int | None = None):
        super().__init__(n_envs, device=device, render=render, n_substeps=n_substeps, max_steps_per_episode=max_steps_per_episode)
        self.mean_distance = self.info_f64[0, :self.n_envs]
        self.reward = torch.zeros(self.n_envs, dtype=torch.float64, device=self.device)   # get_reward is the constant 0 (stacking.py:421-423)
        self._contexts = None

    def reset(self, mask: torch.Tensor | None = None, random: bool = False, context=None):
        """env.reset(random=False, context=...): ``context`` is f64[n_envs, 21] (numpy or tensor; see module docstring).  With
        ``random=True`` contexts are sampled like BlockContextManager.sample.  A mask resets a subset."""
        if context is None:
            if not random and self._contexts is None:
                raise ValueError("CubeStacking_Env.reset needs a context (or random=True)")
            context = sample_contexts(self.n_envs, seed=int(np.random.randint(0, 2 ** 31 - 1))) if random else self._contexts
        ctx = torch.as_tensor(context, dtype=torch.float64).to(self.device).contiguous()
        if tuple(ctx.shape) != (self.n_envs, 21):
            raise ValueError("context must have shape (%d, 21)" % self.n_envs)
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
        """action f64[n, 8] = 7 joint targets + gripper command.  Returns (obs f32[n, 12], reward f64[n] = 0, done u8[n], info) with
        info = dict(mode int16[n] (code, see module docstring), success u8[n], success_1 / success_2 bool[n], mean_distance f64[n])."""
        if action.device != self.device or action.dtype != torch.float64 or tuple(action.shape) != (self.n_envs, 8) or not action.is_contiguous():
            raise ValueError("action must be a contiguous float64 tensor of shape (%d, 8) on %s" % (self.n_envs, self.device))
        with torch.cuda.device(self.device):
            capi.check(self.L.d3il_step(self.h, C.c_void_p(action.data_ptr()), self._stream()))
        n_mode = self.mode & 3
        return self.obs, self.reward, self.done, dict(mode=self.mode, success=self.success, success_1=n_mode > 0, success_2=n_mode > 1,
                                                      mean_distance=self.mean_distance)

    def robot_state(self):
        """CubeStacking_Env.robot_state()[0] (stacking.py:199-212): f64[n, 8] = 7 joint positions + gripper width (finger 1 + finger 2)."""
        q = self.state[0:9, :self.n_envs]
        return torch.cat((q[0:7].t(), (q[7] + q[8]).unsqueeze(1)), dim=1)

    def tcp_pos(self):
        return self.state[capi.STATE_TCP:capi.STATE_TCP + 3, :self.n_envs].t()

    def box_state(self):
        """(pos f64[n, 3, 3], quat f64[n, 3, 4]) of the red, green and blue box."""
        s = self.state[28:28 + 39, :self.n_envs].t().reshape(self.n_envs, 3, 13)
        return s[:, :, 0:3], s[:, :, 3:7]

    def mode_encoding(self):
        return self.mode

    def count_metrics(self, out=None):
        raise capi.D3ilError("count_metrics is Avoiding only; see simulation/stacking_sim.py for the Stacking metrics")

    def policy_action(self, *a, **k):
        raise capi.D3ilError("the device-side random policy drives the Cartesian tasks only")
