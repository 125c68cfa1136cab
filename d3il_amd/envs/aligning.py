"""Batched Aligning environment: mirrors ``Robot_Push_Env``
(environments/d3il/envs/gym_aligning_env/gym_aligning/envs/aligning.py:132-389) for N environments on one GPU.

The rod robot of the Pushing task pushes ONE free compound body - ``robot_push_box.xml``: a 10 x 10 x 2 cm plate of 1 kg (friction 0.3, geom
priority 1) carrying four 1 g walls - from inside or from outside the walls to a target pose.  The target body has sites only: its pose comes
with the context and enters observation, reward and success.  Everything per step runs in libd3il_rollout.so (``k_aligning_step``: variant 2 of the
wave-cooperative engine, d3il_amd/csrc/align_step.h); this class holds no compute.

* observation f32[n, 17] = robot_pos (TCP xyz) | box pos, quat | target pos, quat (aligning.py:223-252; the 8 of ``self.observation_space`` is
  not what ``get_observation`` returns);
* action f64[n, 7] = desired TCP (x, y, z, qw, qx, qy, qz): the harness commands x, y AND z (aligning_sim.py:98-104);
* ``info['mode']`` int16[n]: 0 = the rod is within 5.1 cm of the box centre in xy (inside the walls), 1 = outside (aligning.py:288-312; -1 after a reset);
* success: position error <= 1.8 cm and ``2 arccos|p . q| / pi`` <= 0.048 (aligning.py:325-342); episode cap 400 steps (aligning.py:196);
* contexts f64[n, 14] = box (x, y, z = 0, quat wxyz) | target (x, y, 0, quat) (BlockContextManager.set_context, aligning.py:107-122).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import capi
from .aligning_data import contexts_from_reference, load_test_contexts, sample_contexts  # noqa: F401  (re-exported)
from .avoiding import ObstacleAvoidanceVecEnv


class RobotPushVecEnv(ObstacleAvoidanceVecEnv):
    task = "aligning"
    action_dim = 7
    obs_dim = 17
    default_max_steps = 400          # aligning.py:196

    def __init__(self, n_envs, device=0, render=False, n_substeps: int = 35, max_steps_per_episode: int | None = None):
        super().__init__(n_envs, device=device, render=render, n_substeps=n_substeps, max_steps_per_episode=max_steps_per_episode)
        self.mean_distance = self.info_f64[0, :self.n_envs]
        self.reward = self.info_f64[1, :self.n_envs]
        self._contexts = None

    def reset(self, mask: torch.Tensor | None = None, random: bool = False, context=None):
        """env.reset(random=False, context=...): ``context`` f64[n_envs, 14]; ``random=True`` samples like BlockContextManager.sample
        (aligning.py:59-101).  A mask resets a subset."""
        if context is None:
            if not random and self._contexts is None:
                raise ValueError("Robot_Push_Env.reset needs a context (or random=True)")
            context = sample_contexts(self.n_envs, seed=int(np.random.randint(0, 2 ** 31 - 1))) if random else self._contexts
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
        """Returns (obs f32[n, 17], reward f64[n], done u8[n], info) with info = dict(mode int16[n] in -1..1, success u8[n], mean_distance f64[n]) -
        aligning.py:282-286."""
        if action.device != self.device or action.dtype != torch.float64 or tuple(action.shape) != (self.n_envs, 7) or not action.is_contiguous():
            raise ValueError("action must be a contiguous float64 tensor of shape (%d, 7) on %s" % (self.n_envs, self.device))
        with torch.cuda.device(self.device):
            capi.check(self.L.d3il_step(self.h, C.c_void_p(action.data_ptr()), self._stream()))
        return self.obs, self.reward, self.done, dict(mode=self.mode, success=self.success, mean_distance=self.mean_distance)

    def box_state(self):
        """(pos f64[n, 3], quat f64[n, 4]) of the push box (scene.get_obj_pos / get_obj_quat, MjScene.py:225-247)."""
        s = self.state[capi.ALIGN_STATE_BOX:capi.ALIGN_STATE_BOX + 7, :self.n_envs].t()
        return s[:, 0:3], s[:, 3:7]

    def mode_encoding(self):
        return self.mode

    def count_metrics(self, out=None):
        raise capi.D3ilError("count_metrics is Avoiding only; see simulation/aligning_sim.py for the Aligning metrics")
