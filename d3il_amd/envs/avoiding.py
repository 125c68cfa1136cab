"""Batched counterpart of ``ObstacleAvoidanceEnv``
(environments/d3il/envs/gym_avoiding_env/gym_avoiding/envs/avoiding.py:52-284) over libd3il_rollout.

Same protocol as the reference env (``start() / reset() / step(action) / robot_state()``), but for
``n_envs`` environments at once and with device-resident tensors: observations, done/success flags and
mode codes are zero-copy PyTorch-ROCm views of the library's HBM buffers, so a policy can consume them in
place.  All compute happens in the HIP kernels; this class only moves pointers.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import capi
from ..controllers.offline_ik import offline_ik
from ..kinematics import UrdfChain
from ..model import blob as blob_mod


class _DevArray:
    """Adapter exposing a raw device pointer through __cuda_array_interface__ (zero-copy torch view)."""

    def __init__(self, ptr, shape, typestr, owner):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}
        self._owner = owner


def _view(ptr, shape, typestr, device, owner):
    return torch.as_tensor(_DevArray(ptr, shape, typestr, owner), device=device)


class ObstacleAvoidanceVecEnv:
    task = "avoiding"
    action_dim = 7
    obs_dim = 2

    default_max_steps = 250

    def __init__(self, n_envs: int, device: int | str | torch.device = 0, render: bool = False,
                 n_substeps: int = 35, max_steps_per_episode: int | None = None):
        if max_steps_per_episode is None:
            max_steps_per_episode = self.default_max_steps
        if render:
            raise NotImplementedError("rendering is outside the batched rollout path")
        dev = torch.device(device if not isinstance(device, int) else "cuda:%d" % device)
        if dev.type != "cuda":
            raise capi.D3ilError("%s needs a HIP device (got %s); there is no CPU fallback" % (type(self).__name__, dev))
        self.device = dev
        self.n_envs = int(n_envs)
        self.L = capi.load()
        self.js = blob_mod.load_json(self.task)
        self.js["task_const"]["n_substeps"] = int(n_substeps)
        self.js["task_const"]["max_steps"] = int(max_steps_per_episode)
        self.blob = blob_mod.pack(self.js)
        self.n_substeps, self.max_steps_per_episode = int(n_substeps), int(max_steps_per_episode)
        h = C.c_void_p()
        capi.check(self.L.d3il_create(self.blob.task_id, self.n_envs, dev.index or 0, C.byref(self.blob), C.sizeof(self.blob), C.byref(h)))
        self.h = h
        b = capi.Buffers()
        capi.check(self.L.d3il_get_buffers(self.h, C.byref(b)))
        self.stride = b.stride
        n, s = self.n_envs, b.stride
        self.obs = _view(b.obs, (n, b.obs_dim), "<f4", dev, self)
        self.done = _view(b.done, (n,), "|u1", dev, self)
        self.success = _view(b.success, (n,), "|u1", dev, self)
        self.mode = _view(b.mode, (n,), "<i2", dev, self)            # 9-bit code, fits int16
        self.state_rows = b.state_rows
        self.state = _view(b.state, (b.state_rows, s), "<f8", dev, self)
        self.info_f64 = _view(b.info_f64, (b.n_info_f64, s), "<f8", dev, self) if b.n_info_f64 else None
        self.flags = _view(b.flags, (s,), "<i4", dev, self)
        self.step_count = _view(b.step_count, (s,), "<i4", dev, self)
        self.policy_des = _view(b.policy_des, (3, s), "<f8", dev, self)
        self.last_reset = _view(b.last_reset, (n,), "|u1", dev, self)   # environments reset by the last auto_reset()
        self._tally = None
        self._bound_stream = None
        self.init_qpos = None
        self._started = False

    # ------------------------------------------------------------------ protocol
    def start(self):
        """env.start(): offline IK to the initial TCP pose -> init_qpos (avoiding.py:121-166)."""
        c, tc = self.js["controller"], self.js["task_const"]
        chain = UrdfChain(self.js["urdf_chain"])
        target = list(tc["init_end_eff_pos"]) + list(tc["init_end_eff_quat"])
        q, iters, err = offline_ik(chain, c["default_qpos"], target, np.array(c["joint_pos_min"]), np.array(c["joint_pos_max"]))
        self.set_init_qpos(q)
        return q, iters, err

    def set_init_qpos(self, q):
        q = np.ascontiguousarray(q, dtype=np.float64)
        capi.check(self.L.d3il_start(self.h, q.ctypes.data_as(C.c_void_p)))
        self.init_qpos = q.copy()
        self._started = True

    def _stream(self):
        if self._bound_stream is not None:
            return self._bound_stream
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def bind_stream(self, stream: "torch.cuda.Stream | None"):
        """Launch this environment's kernels on ``stream`` from now on instead of torch's current stream (None: back to the current stream).  For harnesses
        that step several environment batches on their own streams (bench.py --sub-batches) without a stream context per call."""
        self._bound_stream = None if stream is None else C.c_void_p(stream.cuda_stream)

    def step_auto_reset(self, action: torch.Tensor, episode_counts: torch.Tensor):
        """env.step(action) followed by auto_reset(episode_counts) in ONE library call (d3il_step_auto_reset); returns nothing - read obs / done / ... from
        the buffers.  The shape / dtype checks of step() are the caller's business here."""
        capi.check(self.L.d3il_step_auto_reset(self.h, C.c_void_p(action.data_ptr()), C.c_void_p(episode_counts.data_ptr()), self._stream()))

    def random_rollout_step(self, seed: int, env_offset: int, t: int, actions: torch.Tensor, episode_counts: torch.Tensor):
        """policy_action + step + auto_reset of the random-policy harness (BASELINE config 2) in ONE library call (d3il_random_rollout_step)."""
        capi.check(self.L.d3il_random_rollout_step(self.h, int(seed), int(env_offset), int(t), C.c_void_p(actions.data_ptr()), C.c_void_p(episode_counts.data_ptr()), self._stream()))

    def random_rollout_prepare(self, seed: int, env_offset: int, t: int, actions: torch.Tensor, episode_counts: torch.Tensor):
        """With option graph_rollout: capture the HIP graphs the next random_rollout_step calls with these arguments launch (d3il_random_rollout_prepare)."""
        capi.check(self.L.d3il_random_rollout_prepare(self.h, int(seed), int(env_offset), int(t), C.c_void_p(actions.data_ptr()), C.c_void_p(episode_counts.data_ptr()), self._stream()))

    def timing_stats(self):
        """(sum ms, min ms, max ms, launches) of the step-kernel launches since set_timing(True) (d3il_timing_stats; drains the event ring)."""
        out = (C.c_double * 4)()
        capi.check(self.L.d3il_timing_stats(self.h, out))
        return float(out[0]), float(out[1]), float(out[2]), int(out[3])

    def reset(self, mask: torch.Tensor | None = None, random=True, context=None):
        """env.reset() for all environments, or for those with mask != 0 (device uint8/bool tensor).
        The done/success/mode outputs of the reset environments are cleared - pass a copy if ``mask`` is ``self.done``
        and is needed afterwards (e.g. for ``policy_begin``)."""
        mp = None
        if mask is not None:
            mask = mask.to(device=self.device, dtype=torch.uint8).contiguous()
            assert mask.numel() == self.n_envs
            mp = C.c_void_p(mask.data_ptr())
        with torch.cuda.device(self.device):
            capi.check(self.L.d3il_reset(self.h, mp, None, self._stream()))
        return self.obs

    def step(self, action: torch.Tensor):
        """env.step(action): action f64[n_envs, 7] = desired TCP (x, y, z, qw, qx, qy, qz) on this device.
        Returns (obs, reward, done, info) with info = (mode_code, success) like avoiding.py:168-171."""
        if action.device != self.device or action.dtype != torch.float64 or tuple(action.shape) != (self.n_envs, 7) or not action.is_contiguous():
            raise ValueError("action must be a contiguous float64 tensor of shape (%d, 7) on %s" % (self.n_envs, self.device))
        with torch.cuda.device(self.device):
            capi.check(self.L.d3il_step(self.h, C.c_void_p(action.data_ptr()), self._stream()))
        return self.obs, None, self.done, (self.mode, self.success)

    def robot_state(self):
        """env.robot_state(): TCP position f64[n_envs, 3] (gym_env_wrapper.py:160-177)."""
        return self.state[capi.STATE_TCP:capi.STATE_TCP + 3, :self.n_envs].t()

    def mode_encoding(self):
        """mode codes expanded to the reference's float [n_envs, 9] layout."""
        bits = torch.arange(9, device=self.device)
        return ((self.mode.to(torch.int32).unsqueeze(1) >> bits) & 1).to(torch.float32)

    # ------------------------------------------------------------------ extras
    def get_state(self):
        st = np.zeros((self.state_rows, self.n_envs))
        fl = np.zeros(self.n_envs, dtype=np.uint32)
        sc = np.zeros(self.n_envs, dtype=np.int32)
        capi.check(self.L.d3il_get_state(self.h, st.ctypes.data_as(C.c_void_p), st.shape[0], fl.ctypes.data_as(C.c_void_p), sc.ctypes.data_as(C.c_void_p)))
        return st, fl, sc

    def set_state(self, st, fl, sc):
        st = np.ascontiguousarray(st, np.float64); fl = np.ascontiguousarray(fl, np.uint32); sc = np.ascontiguousarray(sc, np.int32)
        assert st.ndim == 2 and st.shape[1] == self.n_envs      # the row count is checked by the library (D3IL_EINVAL on a mismatch)
        capi.check(self.L.d3il_set_state(self.h, st.ctypes.data_as(C.c_void_p), st.shape[0], fl.ctypes.data_as(C.c_void_p), sc.ctypes.data_as(C.c_void_p)))

    def policy_begin(self, mask: torch.Tensor | None = None):
        mp = None
        if mask is not None:
            mask = mask.to(device=self.device, dtype=torch.uint8).contiguous()
            mp = C.c_void_p(mask.data_ptr())
        with torch.cuda.device(self.device):
            capi.check(self.L.d3il_policy_begin(self.h, mp, self._stream()))

    def policy_action(self, seed: int, env_offset: int, t: int, out: torch.Tensor):
        with torch.cuda.device(self.device):
            capi.check(self.L.d3il_policy_action(self.h, int(seed), int(env_offset), int(t), C.c_void_p(out.data_ptr()), self._stream()))
        return out

    def auto_reset(self, episode_counts: torch.Tensor | None = None):
        """Start the next trajectory in every finished lane: reset it (Pushing / Sorting: with the context of its last reset),
        re-latch the harness set-point ``policy_des`` := TCP, episode_counts (int64[2]) += (finished, successes), add the
        episode to the tally table when one is set; ``last_reset`` marks the lanes that were reset.  No host synchronisation."""
        if episode_counts is None:
            if self.task == "avoiding":
                raise ValueError("the Avoiding auto-reset needs episode_counts (int64[2] on the env's device)")
            ptr = None
        else:
            assert episode_counts.dtype == torch.int64 and episode_counts.numel() >= 2 and episode_counts.device == self.device
            ptr = C.c_void_p(episode_counts.data_ptr())
        with torch.cuda.device(self.device):
            capi.check(self.L.d3il_auto_reset(self.h, ptr, self._stream()))

    def set_tally(self, n_ctx: int = 1, ctx_id: torch.Tensor | None = None):
        """Per-context episode tally filled by auto_reset(): returns the int64 [n_ctx, 514] table (episodes, successes, successes by
        mode code; include/d3il_rollout.h).  ctx_id: int32[n_envs] context index of every lane (None: one context)."""
        table = torch.zeros(int(n_ctx), capi.TALLY_ROW, dtype=torch.int64, device=self.device)
        cp = None
        if ctx_id is not None:
            ctx_id = ctx_id.to(device=self.device, dtype=torch.int32).contiguous()
            assert ctx_id.numel() == self.n_envs
            cp = C.c_void_p(ctx_id.data_ptr())
        capi.check(self.L.d3il_set_tally(self.h, cp, int(n_ctx), C.c_void_p(table.data_ptr())))
        self._tally = (table, ctx_id)            # keeps the device memory alive while the library holds the pointers
        return table

    def clear_tally(self):
        capi.check(self.L.d3il_set_tally(self.h, None, 0, None))
        self._tally = None

    def count_metrics(self, out: torch.Tensor | None = None):
        if out is None:
            out = torch.zeros(514, dtype=torch.int64, device=self.device)
        with torch.cuda.device(self.device):
            capi.check(self.L.d3il_count_metrics(self.h, C.c_void_p(out.data_ptr()), self._stream()))
        return out

    def set_timing(self, enabled: bool):
        capi.check(self.L.d3il_set_timing(self.h, int(enabled)))

    def last_step_ms(self) -> float:
        ms = C.c_float()
        capi.check(self.L.d3il_last_step_ms(self.h, C.byref(ms)))
        return float(ms.value)

    def set_option(self, name: str, value: int):
        capi.check(self.L.d3il_set_option(self.h, name.encode(), int(value)))

    def close(self):
        if getattr(self, "h", None):
            self.L.d3il_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
