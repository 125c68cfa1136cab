"""Aligning task (gym_aligning/envs/aligning.py), SURVEY section 8(f)-4: contexts.

The reference's 60 evaluation contexts as data, the reader of its pickle format and a sampler like ``BlockContextManager.sample``.  The environment
itself is ``d3il_amd.envs.aligning.RobotPushVecEnv`` (device engine: d3il_amd/csrc/align_step.h), the evaluation harness
``d3il_amd.simulation.aligning_sim.Aligning_Sim``; task logic and metric tail are pinned against the reference's Python (tests/test_aligning_oracle.py).

Context layout (f64[14], the env format of the other box tasks): box (x, y, z = 0, quat wxyz) | target (x, y, 0, quat) - BlockContextManager.set_context,
aligning.py:107-122."""
from __future__ import annotations

import os

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data")


def _yaw_quat(deg):
    a = np.asarray(deg, dtype=np.float64) * np.pi / 180.0
    out = np.zeros(a.shape + (4,))
    out[..., 0], out[..., 3] = np.cos(a / 2), np.sin(a / 2)
    return out


def load_test_contexts() -> np.ndarray:
    """The reference's 60 evaluation contexts (environments/dataset/data/aligning/test_contexts.pkl) as f64[60, 14]."""
    return np.load(os.path.join(_DATA, "aligning_test_contexts.npy"))


def contexts_from_reference(ctx_list) -> np.ndarray:
    """The reference's pickle format - a list of [pos (x, y, yaw deg), quat, target_pos (x, y, yaw deg), target_quat] - as f64[n, 14]."""
    out = np.zeros((len(ctx_list), 14))
    for i, (pos, quat, tpos, tquat) in enumerate(ctx_list):
        out[i, 0:2], out[i, 3:7], out[i, 7:9], out[i, 10:14] = pos[:2], quat, tpos[:2], tquat
    return out


def sample_contexts(n: int, seed: int = 0) -> np.ndarray:
    """Contexts drawn like BlockContextManager.sample (aligning.py:59-101): box x in [0.4, 0.6], y in [-0.25, -0.1], target x in [0.4, 0.6], y in
    [0.2, 0.35], yaw in [-90, 90] degrees each (gym Box spaces, float32)."""
    rng = np.random.default_rng(seed)
    box = rng.uniform([0.4, -0.25, -90], [0.6, -0.1, 90], size=(n, 3)).astype(np.float32).astype(np.float64)
    tgt = rng.uniform([0.4, 0.2, -90], [0.6, 0.35, 90], size=(n, 3)).astype(np.float32).astype(np.float64)
    out = np.zeros((n, 14))
    out[:, 0:2], out[:, 3:7] = box[:, :2], _yaw_quat(box[:, 2])
    out[:, 7:9], out[:, 10:14] = tgt[:, :2], _yaw_quat(tgt[:, 2])
    return out
