"""Host-side (cold path) kinematics of the controller's URDF chain, in numpy.

Counterpart of ``RobotModelFromPinochio.getForwardKinematics/getJacobian``
(environments/d3il/d3il_sim/core/Model.py:37-66).  pinocchio itself is an un-vendored
dependency; what it computes for a serial chain of revolute joints is standard:
frame placement = product of (fixed joint placement, rotation about the joint axis), and the
LOCAL_WORLD_ALIGNED frame Jacobian has columns ``[z_i x (p - o_i); z_i]``.  The quaternion is
extracted with Eigen's matrix->quaternion branch rule, which is what
``pinocchio.Quaternion(R)`` runs (Model.py:47-53) [ext].

Used only by ``env.start()`` (offline IK, once per environment set) and by the golden-vector
generator; the per-step kinematics run inside the HIP kernels.
"""
from __future__ import annotations

import math

import numpy as np


def mat2quat_eigen(R):
    t = R[0, 0] + R[1, 1] + R[2, 2]
    q = np.zeros(4)
    if t > 0:
        t = math.sqrt(t + 1.0)
        q[0] = 0.5 * t
        t = 0.5 / t
        q[1] = (R[2, 1] - R[1, 2]) * t
        q[2] = (R[0, 2] - R[2, 0]) * t
        q[3] = (R[1, 0] - R[0, 1]) * t
    else:
        i = 0
        if R[1, 1] > R[0, 0]:
            i = 1
        if R[2, 2] > R[i, i]:
            i = 2
        j = (i + 1) % 3
        k = (j + 1) % 3
        t = math.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
        q[1 + i] = 0.5 * t
        t = 0.5 / t
        q[0] = (R[k, j] - R[j, k]) * t
        q[1 + j] = (R[j, i] + R[i, j]) * t
        q[1 + k] = (R[k, i] + R[i, k]) * t
    return q


class UrdfChain:
    """Serial chain described by the ``urdf_chain`` section of the model blob JSON."""

    def __init__(self, chain):
        self.links = [(np.asarray(c["xyz"], float), np.asarray(c["R"], float),
                       c["type"] == "revolute", np.asarray(c["axis"], float)) for c in chain]
        self.ndof = sum(1 for l in self.links if l[2])

    def _walk(self, q):
        R = np.eye(3)
        p = np.zeros(3)
        axes, origins = [], []
        k = 0
        for xyz, Rf, rev, axis in self.links:
            p = p + R @ xyz
            R = R @ Rf
            if rev:
                c, s = math.cos(q[k]), math.sin(q[k])
                x, y, z = axis
                K = np.array([[0, -z, y], [z, 0, -x], [-y, x, 0]])
                Rq = np.eye(3) + s * K + (1 - c) * (K @ K)
                axes.append(R @ axis)
                origins.append(p.copy())
                R = R @ Rq
                k += 1
        return p, R, axes, origins

    def fk(self, q):
        p, R, _, _ = self._walk(q)
        return p, mat2quat_eigen(R)

    def jacobian(self, q):
        p, _, axes, origins = self._walk(q)
        J = np.zeros((6, self.ndof))
        for i, (z, o) in enumerate(zip(axes, origins)):
            J[:3, i] = np.cross(z, p - o)
            J[3:, i] = z
        return J
