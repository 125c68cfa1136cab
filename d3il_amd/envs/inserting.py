"""Batched counterpart of ``Gate_Insertion_Env`` (environments/d3il/envs/gym_inserting_env/gym_inserting/envs/gate_insertion.py:157-514)
over libd3il_rollout.

Protocol of the reference env - ``start()``, ``reset(random=False, context=...)``, ``step(action)`` returning ``(obs, reward, done, info)``
with ``info = {'success', 'mean_distance', 'mode', 'one_box_success', 'two_box_success', 'three_box_success'}`` (:386-409), ``robot_state()`` -
for ``n_envs`` environments at once, all tensors device resident (zero-copy views of the library's HBM buffers).  The task runs on the
generic engine of the Sorting task (csrc/gen_step.h: three 5 cm cubes, the seventeen static walls of the three gates, and - because the rod
works between those walls - the rod <-> wall contacts).

Contexts.  ``BlockContextManager.sample`` (:70-87) draws (x, y, yaw in degrees) for the three push boxes from three gym Box spaces and
``set_context`` (:89-113) places each at ``[x, y, 0.0]`` with the yaw quaternion; here a context is the f64 row ``(x, y, 0, qw, qx, qy, qz) x 3`` in
the order push_box1 (red), push_box2 (green), push_box3 (blue).

The observation has 11 entries (TCP xy, then x, y, tan(yaw) per box, :278-309); the reference declares an observation space of 14 but returns 11.
The reference ships neither a config nor a Sim class nor a dataset for this task, so there is no ``Inserting_Sim`` here either.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import capi
from .avoiding import ObstacleAvoidanceVecEnv
from .pushing import _yaw_quat

_SPACES = np.array([[0.35, -0.2, 0.5, -0.15], [0.55, -0.1, 0.7, -0.05], [0.35, 0.0, 0.5, 0.05]])   # gate_insertion.py:53-63 (x lo, y lo, x hi, y hi)
MODE_DICT = {"rgb": 1, "rbg": 2, "grb": 3, "gbr": 4, "brg": 5, "bgr": 6}                        # gate_insertion.py:283


def contexts_from_reference(ctx_list) -> np.ndarray:
    """[[pos1 (x, y, deg), quat1, pos2, quat2, pos3, quat3], ...] (BlockContextManager.sample) -> f64 [n, 21]."""
    out = np.zeros((len(ctx_list), 3, 7))
    for i, c in enumerate(ctx_list):
        for k in range(3):
            out[i, k, 0:2] = np.asarray(c[2 * k], dtype=np.float64)[:2]
            out[i, k, 3:7] = np.asarray(c[2 * k + 1], dtype=np.float64)
    return out.reshape(len(ctx_list), 21)


def sample_contexts(n: int, seed: int = 0) -> np.ndarray:
    """Contexts drawn like BlockContextManager.sample: (x, y) from the three boxes' spaces, yaw in [-90, 90] degrees (gym Box spaces: float32)."""
    rng = np.random.default_rng(seed)
    out = np.zeros((n, 3, 7))
    for i in range(n):
        xy = rng.uniform(_SPACES[:, :2], _SPACES[:, 2:]).astype(np.float32).astype(np.float64)
        yaw = rng.uniform(-90, 90, size=3).astype(np.float32).astype(np.float64)
        out[i, :, 0:2], out[i, :, 3:7] = xy, _yaw_quat(yaw)
    return out.reshape(n, 21)


class GateInsertionVecEnv(ObstacleAvoidanceVecEnv):
    task = "inserting"
    action_dim = 7
    obs_dim = 11
    default_max_steps = 2000          # gate_insertion.py:158

    def __init__(self, n_envs, device=0, render=False, n_substeps: int = 35, max_steps_per_episode: int | None = None):
        super().__init__(n_envs, device=device, render=render, n_substeps=n_substeps, max_steps_per_episode=max_steps_per_episode)
        self._contexts = None
        self.box_row, self.warm_row, self.task_row = capi.INS_STATE_BOX, capi.INS_STATE_WARM, capi.INS_STATE_TASK
        self.targets = torch.as_tensor(np.asarray(self.js["task_const"]["target_pos"], dtype=np.float64), device=self.device)

    def reset(self, mask: torch.Tensor | None = None, random: bool = False, context=None):
        """env.reset(random=False, context=...): ``context`` is f64[n_envs, 21] (numpy or tensor; see module docstring).  With ``random=True`` contexts are
        sampled like BlockContextManager.sample.  A mask resets a subset."""
        if context is None:
            if not random and self._contexts is None:
                raise ValueError("Gate_Insertion_Env.reset needs a context (or random=True)")
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
        """Returns (obs f32[n, 11], reward f64[n], done u8[n], info) - gate_insertion.py:386-409; info['mode'] is the mode_dict code (0 until all three
        boxes have been in their goals), the n-box successes are the number of letters in the env's ``modes`` list >= n."""
        if action.device != self.device or action.dtype != torch.float64 or tuple(action.shape) != (self.n_envs, 7) or not action.is_contiguous():
            raise ValueError("action must be a contiguous float64 tensor of shape (%d, 7) on %s" % (self.n_envs, self.device))
        # obs / reward / done are sampled BEFORE the sub-steps (GymEnvWrapper.step, gym_env_wrapper.py:88-93): the reward of this call belongs to
        # the state the step starts from, like the obs / done the kernel writes (sort_step_begin)
        reward = self.get_reward()
        with torch.cuda.device(self.device):
            capi.check(self.L.d3il_step(self.h, C.c_void_p(action.data_ptr()), self._stream()))
        code = self.mode.to(torch.int32)
        nm = code >> 3
        info = dict(success=self.success, mean_distance=self.state[self.task_row + 1, :self.n_envs], mode=code & 7,
                    one_box_success=(nm >= 1).to(torch.uint8), two_box_success=(nm >= 2).to(torch.uint8), three_box_success=(nm >= 3).to(torch.uint8))
        return self.obs, reward, self.done, info

    def get_reward(self):
        """-(min distance robot <-> box in xy + the three 3-D box <-> target distances) (gate_insertion.py:448-473), from the state buffer."""
        pos, _ = self.box_state()
        tcp = self.robot_state()[:, :2]
        dr = (pos[:, :, :2] - tcp.unsqueeze(1)).norm(dim=2).min(dim=1).values
        dt = (pos - self.targets.unsqueeze(0)).norm(dim=2).sum(dim=1)
        return -(dr + dt)

    def box_state(self):
        """(pos f64[n, 3, 3], quat f64[n, 3, 4]) of push_box1..3 (scene.get_obj_pos / get_obj_quat)."""
        s = self.state[self.box_row:self.box_row + 39, :self.n_envs].t().reshape(self.n_envs, 3, 13)
        return s[:, :, 0:3], s[:, :, 3:7]

    def mode_letters(self):
        """The env's ``modes`` list per environment as strings ('', 'g', 'gb', 'gbr', ...), from the task word of the state buffer."""
        w = self.state[self.task_row, :self.n_envs].to(torch.int64).cpu().numpy()
        return ["".join("rgb"[((int(x) >> (2 + 2 * k)) & 3) - 1] for k in range(int(x) & 3)) for x in w]

    def mode_encoding(self):
        return self.mode

    def count_metrics(self, out=None):
        raise capi.D3ilError("count_metrics is Avoiding only; the reference defines no metrics for the Inserting task")
