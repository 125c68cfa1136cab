"""Scripted pick-and-place trajectories for the Stacking task (measurement / test harness, not part of the reference).

The Stacking action is joint space (7 joint targets + gripper command, stacking.py:331-346), so a scripted policy needs inverse
kinematics: the Cartesian way-points of "grasp box b, carry it to the target zone, put it down at stack level k" are converted
once per context on the host with the same damped least-squares iteration as the offline IK of env.start()
(controllers/offline_ik.py) and replayed as a table of per-step actions.  Used by bench.py --task stacking, the Stacking tests and
tools/: it exercises every contact class of the task (finger-tip and finger-hull grasp contacts, box on box, box on table).
"""
from __future__ import annotations

import numpy as np

from ..kinematics import UrdfChain
from .offline_ik import offline_ik

BOX_HALF = np.array([[0.03, 0.03, 0.03], [0.03, 0.03, 0.03], [0.03, 0.05, 0.03]])     # stacking_objects.py:21-50
REST_Z = 0.011            # centre height of a 6 cm box resting on the table top (table surface at z = -0.019)
TARGET = np.array([0.5, 0.2])


def _yaw_of(quat):
    w, x, y, z = quat
    return np.arctan2(2 * (w * z + x * y), 1 - 2 * (y * y + z * z))


def _grip_quat(phi):
    """TCP orientation Rz(phi) * [0, 1, 0, 0]: gripper pointing down, closing axis turned by phi about the vertical."""
    return np.array([0.0, np.cos(phi / 2), np.sin(phi / 2), 0.0])


def grasp_yaw(box: int, quat) -> float:
    """Yaw of the gripper for box ``box`` with orientation ``quat``: the fingers close along the hand's y axis, which must line up
    with a 6 cm extent of the box - any face pair of a cube (yaw modulo 90 degrees), the x extent of the 6 x 10 x 6 box."""
    psi = _yaw_of(quat)
    if box < 2:
        return (psi + np.pi / 4) % (np.pi / 2) - np.pi / 4
    phi = psi + np.pi / 2
    return (phi + np.pi / 2) % np.pi - np.pi / 2


def build_trajectory(js: dict, init_qpos, ctx21, order=(0, 1, 2), speed: float = 1.0, n_boxes: int | None = None) -> np.ndarray:
    """Per-step actions f64 [T, 8] for one context: for every box in ``order`` - above the box, down, close, up, over the target, down
    to its stack level, open, up.  ``speed`` scales the number of env steps per segment (1.0: ~95 steps per box)."""
    chain = UrdfChain(js["urdf_chain"])
    c = js["controller"]
    qmin, qmax = np.array(c["joint_pos_min"]), np.array(c["joint_pos_max"])
    ctx = np.asarray(ctx21, dtype=np.float64).reshape(3, 7)
    q = np.asarray(init_qpos, dtype=np.float64).copy()
    acts = []

    def seg(pos, phi, grip, n):
        nonlocal q
        n = max(2, int(round(n / speed)))
        qt = offline_ik(chain, q, list(pos) + list(_grip_quat(phi)), qmin, qmax, eps=1e-8, it_max=600)[0]
        for k in range(n):
            a = q + (qt - q) * min(1.0, (k + 1) / (0.7 * n))
            acts.append(np.concatenate([a, [grip]]))
        q = qt

    boxes = list(order)[: (len(order) if n_boxes is None else n_boxes)]
    for level, b in enumerate(boxes):
        x, y = ctx[b, 0], ctx[b, 1]
        phi = grasp_yaw(b, ctx[b, 3:7])
        z_pick = REST_Z + 0.004
        z_place = REST_Z + 0.06 * level + 0.006
        seg([x, y, 0.16], phi, 1.0, 26)
        seg([x, y, z_pick + 0.05], phi, 1.0, 12)
        seg([x, y, z_pick], phi, 1.0, 14)
        seg([x, y, z_pick], phi, 0.0, 8)                       # close
        seg([x, y, 0.10 + 0.06 * level], phi, 0.0, 16)
        seg([TARGET[0], TARGET[1], 0.12 + 0.06 * level], 0.0, 0.0, 30)
        seg([TARGET[0], TARGET[1], z_place + 0.03], 0.0, 0.0, 12)
        seg([TARGET[0], TARGET[1], z_place], 0.0, 0.0, 12)
        seg([TARGET[0], TARGET[1], z_place], 0.0, 1.0, 6)      # open
        seg([TARGET[0], TARGET[1], z_place + 0.10], 0.0, 1.0, 12)
    return np.array(acts)


def build_palm_press(js: dict, init_qpos, ctx21, box: int = 0, z_low: float = -0.004) -> np.ndarray:
    """Per-step actions f64 [T, 8] that press the PALM of the open gripper onto box ``box``: the fingers straddle the box and the TCP is sent
    to ``z_low`` (the palm, 3.9 cm above the TCP, meets the 6 cm box top at TCP z = 0.002), hold, retreat.  Exercises the box <-> hand-hull
    pair (panda_invisible.xml:72, mesh handv) that no pick-and-place reaches."""
    chain = UrdfChain(js["urdf_chain"])
    c = js["controller"]
    qmin, qmax = np.array(c["joint_pos_min"]), np.array(c["joint_pos_max"])
    ctx = np.asarray(ctx21, dtype=np.float64).reshape(3, 7)
    x, y = ctx[box, 0], ctx[box, 1]
    phi = grasp_yaw(box, ctx[box, 3:7])
    q = np.asarray(init_qpos, dtype=np.float64).copy()
    acts = []
    for pos, n in (([x, y, 0.16], 30), ([x, y, 0.06], 20), ([x, y, z_low], 30), ([x, y, z_low], 15), ([x, y, 0.12], 20)):
        qt = offline_ik(chain, q, list(pos) + list(_grip_quat(phi)), qmin, qmax, eps=1e-8, it_max=600)[0]
        for k in range(n):
            acts.append(np.concatenate([q + (qt - q) * min(1.0, (k + 1) / (0.7 * n)), [1.0]]))
        q = qt
    return np.array(acts)
