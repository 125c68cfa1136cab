"""ctypes loader for the CPU oracle (oracle/d3il_oracle.c).  TEST INFRASTRUCTURE ONLY.

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg - never by
the product package ``d3il_amd``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libd3il_oracle.so")
    src = os.path.join(_HERE, "d3il_oracle.c")
    hdr = os.path.join(os.path.dirname(_HERE), "include", "d3il_model_blob.h")
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libd3il_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_create.restype = C.c_void_p
        for name in ("orc_nq", "orc_nv", "orc_get_contacts", "orc_get_efc", "orc_solver_iter",
                     "orc_unsupported_pairs", "orc_supported_pairs", "orc_test_box_box", "orc_test_cyl_box"):
            getattr(_LIB, name).restype = C.c_int
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def box_box(p1, q1, s1, p2, q2, s2, margin=0.0):
    """Collision test hook: contacts [n][7] = dist, pos3, normal3 (normal from box 1 to box 2)."""
    a = [np.ascontiguousarray(x, float) for x in (p1, q1, s1, p2, q2, s2)]
    out = np.zeros((16, 7))
    n = lib().orc_test_box_box(*[_p(x) for x in a], C.c_double(margin), _p(out))
    return out[:n]


def cyl_box(pc, qc, rad, half, pb, qb, sb, margin=0.0):
    """Collision test hook: [dist, pos3, normal3] (normal from the box to the cylinder) or None."""
    a = [np.ascontiguousarray(x, float) for x in (pc, qc, pb, qb, sb)]
    out = np.zeros(7)
    n = lib().orc_test_cyl_box(_p(a[0]), _p(a[1]), C.c_double(rad), C.c_double(half), _p(a[2]), _p(a[3]), _p(a[4]),
                               C.c_double(margin), _p(out))
    return out if n else None


class Oracle:
    """One scalar environment."""

    def __init__(self, blob):
        self.L = lib()
        self.blob = blob
        self.h = C.c_void_p(self.L.orc_create(C.byref(blob)))
        if not self.h:
            raise RuntimeError("orc_create failed (blob magic/version)")
        self.nq, self.nv = self.L.orc_nq(self.h), self.L.orc_nv(self.h)

    def __del__(self):
        try:
            self.L.orc_destroy(self.h)
        except Exception:
            pass

    # ---- physics level
    def set_state(self, qpos, qvel):
        qpos, qvel = np.ascontiguousarray(qpos, float), np.ascontiguousarray(qvel, float)
        self.L.orc_set_qpos_qvel(self.h, _p(qpos), _p(qvel))

    def set_gravity(self, g):
        g = np.ascontiguousarray(g, float)
        self.L.orc_set_gravity(self.h, _p(g))

    def set_ctrl(self, ctrl):
        ctrl = np.ascontiguousarray(ctrl, float)
        self.L.orc_set_ctrl(self.h, _p(ctrl))

    def forward(self):
        self.L.orc_forward(self.h)

    def mj_step(self):
        self.L.orc_mj_step(self.h)

    def state(self):
        qpos, qvel = np.zeros(self.nq), np.zeros(self.nv)
        self.L.orc_get_qpos_qvel(self.h, _p(qpos), _p(qvel))
        return qpos, qvel

    def M(self):
        M = np.zeros((self.nv, self.nv))
        self.L.orc_get_M(self.h, _p(M))
        return M

    def vec(self, name):
        which = {"qfrc_bias": 0, "qacc": 1, "qacc_smooth": 2, "qfrc_constraint": 3, "qfrc_actuator": 4,
                 "dof_invweight0": 5}[name]
        out = np.zeros(self.nv)
        self.L.orc_get_vec(self.h, which, _p(out))
        return out

    def body(self, b):
        xpos, xquat, invw = np.zeros(3), np.zeros(4), np.zeros(2)
        self.L.orc_get_body(self.h, int(b), _p(xpos), _p(xquat), _p(invw))
        return xpos, xquat, invw

    def contacts(self):
        out = np.zeros((96, 10))
        n = self.L.orc_get_contacts(self.h, _p(out))
        return out[:n]

    def efc(self):
        f, a, d = np.zeros(320), np.zeros(320), np.zeros(320)
        n = self.L.orc_get_efc(self.h, _p(f), _p(a), _p(d))
        return f[:n], a[:n], d[:n]

    def solver_iter(self):
        return self.L.orc_solver_iter(self.h)

    def grad_at(self, x):
        """Optimality residual (gradient of the constraint problem of the last forward pass) at acceleration x."""
        x = np.ascontiguousarray(x, float)
        g = np.zeros(self.nv)
        self.L.orc_grad_at(self.h, _p(x), _p(g))
        return g

    def pairs(self, supported=True):
        out = np.zeros((1024, 2), dtype=np.int32)
        fn = self.L.orc_supported_pairs if supported else self.L.orc_unsupported_pairs
        n = fn(self.h, _p(out), 1024)
        return out[:min(n, 1024)]

    # ---- controller level
    def fk(self, q):
        q = np.ascontiguousarray(q, float)
        pos, quat = np.zeros(3), np.zeros(4)
        self.L.orc_fk(self.h, _p(q), _p(pos), _p(quat))
        return pos, quat

    def jac(self, q):
        q = np.ascontiguousarray(q, float)
        J = np.zeros((6, 7))
        self.L.orc_jac(self.h, _p(q), _p(J))
        return J

    def ik_reset(self):
        self.L.orc_ik_reset(self.h)

    def ik_setpoint(self, a):
        a = np.ascontiguousarray(a, float)
        self.L.orc_ik_setpoint(self.h, _p(a))

    def set_robot_state(self, jpos, jvel, fpos=None, fvel=None, setw=0.001, grasp=False):
        jpos, jvel = np.ascontiguousarray(jpos, float), np.ascontiguousarray(jvel, float)
        if fpos is None:
            self.L.orc_set_robot_state(self.h, _p(jpos), _p(jvel), None, None, C.c_double(setw), int(grasp))
        else:
            fpos, fvel = np.ascontiguousarray(fpos, float), np.ascontiguousarray(fvel, float)
            self.L.orc_set_robot_state(self.h, _p(jpos), _p(jvel), _p(fpos), _p(fvel), C.c_double(setw), int(grasp))

    def ik_control(self):
        tau = np.zeros(7)
        self.L.orc_ik_control(self.h, _p(tau))
        return tau

    def ik_state(self):
        a, b = np.zeros(7), np.zeros(7)
        self.L.orc_get_ik_state(self.h, _p(a), _p(b))
        return a, b

    def pd_control(self, q_des, qd_des):
        q_des, qd_des = np.ascontiguousarray(q_des, float), np.ascontiguousarray(qd_des, float)
        self.L.orc_pd_setpoint(self.h, _p(q_des), _p(qd_des))
        tau = np.zeros(7)
        self.L.orc_pd_control(self.h, _p(tau))
        return tau

    def finger_ctrl(self):
        f = np.zeros(2)
        self.L.orc_finger_ctrl(self.h, _p(f))
        return f

    def check_mode(self, cpos, reset=False):
        cpos = np.ascontiguousarray(cpos, float)
        mode = np.zeros(9)
        succ = C.c_int(0)
        self.L.orc_check_mode(self.h, _p(cpos), int(reset), _p(mode), C.byref(succ))
        return mode, bool(succ.value)

    # ---- env level (Avoiding)
    def env_start(self, init_qpos):
        init_qpos = np.ascontiguousarray(init_qpos, float)
        self.L.orc_env_start(self.h, _p(init_qpos))

    def env_reset(self):
        obs = np.zeros(2, dtype=np.float32)
        self.L.orc_env_reset(self.h, _p(obs))
        return obs

    def env_step(self, action):
        action = np.ascontiguousarray(action, float)
        obs = np.zeros(2, dtype=np.float32)
        mode = np.zeros(9)
        done, succ = C.c_int(0), C.c_int(0)
        self.L.orc_env_step(self.h, _p(action), _p(obs), C.byref(done), _p(mode), C.byref(succ))
        return obs, bool(done.value), mode, bool(succ.value)

    def env_set_state(self, state42, flags, step):
        state42 = np.ascontiguousarray(state42, float)
        self.L.orc_env_set_state(self.h, _p(state42), C.c_uint(int(flags)), int(step))

    def env_state(self):
        s = np.zeros(42)
        self.L.orc_env_get_state(self.h, _p(s))
        f = np.zeros(8, dtype=np.int32)
        self.L.orc_env_get_flags(self.h, _p(f))
        return s, f

    # ---- env level (Pushing)
    def push_reset(self, ctx):
        ctx = np.ascontiguousarray(ctx, float).reshape(14)
        obs = np.zeros(8, dtype=np.float32)
        self.L.orc_push_reset(self.h, _p(ctx), _p(obs))
        return obs

    def push_step(self, action):
        action = np.ascontiguousarray(action, float)
        obs = np.zeros(8, dtype=np.float32)
        done, mode, succ = C.c_int(0), C.c_int(0), C.c_int(0)
        rew, md = C.c_double(0), C.c_double(0)
        self.L.orc_push_step(self.h, _p(action), _p(obs), C.byref(done), C.byref(rew), C.byref(mode), C.byref(succ), C.byref(md))
        return obs, rew.value, bool(done.value), dict(mode=mode.value, success=bool(succ.value), mean_distance=md.value)

    def push_logic(self, box14, tcp, reset=False):
        box14 = np.ascontiguousarray(box14, float).reshape(14)
        tcp = np.ascontiguousarray(tcp, float)
        obs = np.zeros(8, dtype=np.float32)
        succ, mode, first = C.c_int(0), C.c_int(0), C.c_int(0)
        md, rew = C.c_double(0), C.c_double(0)
        self.L.orc_push_logic(self.h, _p(box14), _p(tcp), int(reset), _p(obs), C.byref(succ), C.byref(mode), C.byref(first),
                              C.byref(md), C.byref(rew))
        return obs, bool(succ.value), mode.value, first.value, md.value, rew.value

    # ---- env level (Aligning; oracle only so far - DESIGN section 17.8)
    def align_reset(self, ctx):
        """ctx f64[14]: box (x, y, 0, quat) | target (x, y, 0, quat) -> obs f32[17] (robot xyz | box pos, quat | target pos, quat)."""
        ctx = np.ascontiguousarray(ctx, float).reshape(14)
        obs = np.zeros(17, dtype=np.float32)
        self.L.orc_align_reset(self.h, _p(ctx), _p(obs))
        return obs

    def align_step(self, action):
        action = np.ascontiguousarray(action, float)
        obs = np.zeros(17, dtype=np.float32)
        done, mode, succ = C.c_int(0), C.c_int(0), C.c_int(0)
        rew, md = C.c_double(0), C.c_double(0)
        self.L.orc_align_step(self.h, _p(action), _p(obs), C.byref(done), C.byref(rew), C.byref(mode), C.byref(succ), C.byref(md))
        return obs, rew.value, bool(done.value), dict(mode=mode.value, success=bool(succ.value), mean_distance=md.value)

    def align_set_state(self, s77, step=0, terminated=False, ik_valid=True):
        """Load one environment's column of the HIP path's Aligning state buffer (+ step counter, flags)."""
        s77 = np.ascontiguousarray(s77, float)
        self.L.orc_align_set_state(self.h, _p(s77), int(step), int(terminated), int(ik_valid))

    def align_state(self):
        """(arm q[9] v[9], box pos3 quat4 vel6) of the oracle in the device layout."""
        qp, qv = self.state()
        return np.concatenate([qp[7:16], qv[6:15]]), np.concatenate([qp[0:7], qv[0:6]])

    def align_logic(self, box7, target7, tcp):
        box7, target7, tcp = (np.ascontiguousarray(x, float) for x in (box7, target7, tcp))
        obs = np.zeros(17, dtype=np.float32)
        succ, mode = C.c_int(0), C.c_int(0)
        md, rew = C.c_double(0), C.c_double(0)
        self.L.orc_align_logic(self.h, _p(box7), _p(target7), _p(tcp), _p(obs), C.byref(succ), C.byref(mode), C.byref(md), C.byref(rew))
        return obs, bool(succ.value), mode.value, md.value, rew.value

    # ---- env level (Sorting; oracle only so far)
    def sort_reset(self, ctx):
        ctx = np.ascontiguousarray(ctx, float).reshape(-1)
        obs = np.zeros(2 + 3 * (len(ctx) // 7), dtype=np.float32)
        self.L.orc_sortenv_reset(self.h, _p(ctx), _p(obs))
        return obs

    def sort_step(self, action):
        action = np.ascontiguousarray(action, float)
        obs = np.zeros(2 + 3 * self.blob.n_obj, dtype=np.float32)
        done, code, succ = C.c_int(0), C.c_int(0), C.c_int(0)
        self.L.orc_sortenv_step(self.h, _p(action), _p(obs), C.byref(done), C.byref(code), C.byref(succ))
        return obs, bool(done.value), dict(mode=code.value, success=bool(succ.value))

    def sort_set_state(self, state_col, flags, step):
        """Load one environment's column of the HIP path's Sorting state buffer (+ flags word, step counter)."""
        state_col = np.ascontiguousarray(state_col, float)
        self.L.orc_sort_set_state(self.h, _p(state_col), C.c_uint(int(flags) & 0xFFFFFFFF), int(step))

    # ---- env level (Inserting)
    def ins_reset(self, ctx):
        ctx = np.ascontiguousarray(ctx, float).reshape(21)
        obs = np.zeros(11, dtype=np.float32)
        self.L.orc_insenv_reset(self.h, _p(ctx), _p(obs))
        return obs

    def ins_step(self, action):
        action = np.ascontiguousarray(action, float)
        obs = np.zeros(11, dtype=np.float32)
        done, code, nm, succ, md = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0), C.c_double(0)
        self.L.orc_insenv_step(self.h, _p(action), _p(obs), C.byref(done), C.byref(code), C.byref(nm), C.byref(succ), C.byref(md))
        return obs, bool(done.value), dict(mode=code.value, n_mode=nm.value, success=bool(succ.value), mean_distance=md.value)

    def ins_set_state(self, state_col, flags, step):
        """Load one environment's column of the HIP path's Inserting state buffer (+ flags word, step counter)."""
        state_col = np.ascontiguousarray(state_col, float)
        self.L.orc_ins_set_state(self.h, _p(state_col), C.c_uint(int(flags) & 0xFFFFFFFF), int(step))

    # ---- env level (Stacking)
    def stack_reset(self, ctx):
        """ctx: 3 x (x, y, z = 0, quat) red, green, blue (BlockContextManager.set_context, stacking.py:99-125)."""
        ctx = np.ascontiguousarray(ctx, float).reshape(21)
        obs = np.zeros(12, dtype=np.float32)
        self.L.orc_stackenv_reset(self.h, _p(ctx), _p(obs))
        return obs

    def stack_step(self, action8):
        action8 = np.ascontiguousarray(action8, float).reshape(8)
        obs = np.zeros(12, dtype=np.float32)
        done, succ = C.c_int(0), C.c_int(0)
        md = C.c_double(0)
        mode = C.create_string_buffer(4)
        self.L.orc_stackenv_step(self.h, _p(action8), _p(obs), C.byref(done), mode, C.byref(succ), C.byref(md))
        m = mode.value.decode()
        return obs, bool(done.value), dict(mode=m, success=bool(succ.value), success_1=len(m) > 0, success_2=len(m) > 1, mean_distance=md.value)

    def stack_robot_state(self):
        out = np.zeros(8)
        self.L.orc_stackenv_robot_state(self.h, _p(out))
        return out

    def stack_state(self):
        s = np.zeros(28 + 13 * 3)
        self.L.orc_stack_get_state(self.h, _p(s))
        return s

    def stack_set_state(self, s67, step=0, terminated=False, min_inds=()):
        s67 = np.ascontiguousarray(s67, float)
        mi = np.ascontiguousarray(list(min_inds) + [0] * (3 - len(min_inds)), dtype=np.int32)
        self.L.orc_stack_set_state(self.h, _p(s67), int(step), int(terminated), len(min_inds), _p(mi))

    def push_state(self):
        s = np.zeros(42 + 13 * 2)
        self.L.orc_push_get_state(self.h, _p(s))
        f = np.zeros(8, dtype=np.int32)
        self.L.orc_push_get_flags(self.h, _p(f))
        return s, f

    def push_set_state(self, s68, step=0, terminated=False, first_visit=-1, ik_valid=True):
        s68 = np.ascontiguousarray(s68, float)
        self.L.orc_push_set_state(self.h, _p(s68), int(step), int(terminated), int(first_visit), int(ik_valid))


class SortLogic:
    """Sorting task logic with injected poses (oracle/d3il_oracle.c orc_sort_logic)."""

    def __init__(self, num_boxes):
        self.L = lib()
        self.st = (C.c_int * 14)()
        self.num_boxes = num_boxes
        self.reset()

    def reset(self):
        self.L.orc_sort_reset(self.st)

    def step(self, box42, tcp):
        box42, tcp = np.ascontiguousarray(box42, float).reshape(42), np.ascontiguousarray(tcp, float)
        obs = np.zeros(2 + 3 * self.num_boxes, dtype=np.float32)
        succ, code = C.c_int(0), C.c_int(0)
        self.L.orc_sort_logic(self.st, _p(box42), _p(tcp), self.num_boxes, _p(obs), C.byref(succ), C.byref(code))
        return obs, bool(succ.value), code.value


class InsertLogic:
    """Inserting task logic with injected poses (oracle/d3il_oracle.c orc_ins_logic)."""

    def __init__(self, targets, min_dist):
        self.L = lib()
        self.st = (C.c_int * 2)()
        self.targets = np.ascontiguousarray(targets, float).reshape(9)
        self.min_dist = float(min_dist)
        self.reset()

    def reset(self):
        self.L.orc_ins_reset(self.st)

    def step(self, box21, tcp):
        box21, tcp = np.ascontiguousarray(box21, float).reshape(21), np.ascontiguousarray(tcp, float)
        obs = np.zeros(11, dtype=np.float32)
        succ, nm, code, md = C.c_int(0), C.c_int(0), C.c_int(0), C.c_double(0)
        self.L.orc_ins_logic(self.st, _p(box21), _p(self.targets), C.c_double(self.min_dist), _p(tcp), _p(obs), C.byref(succ), C.byref(md), C.byref(nm), C.byref(code))
        return obs, bool(succ.value), md.value, nm.value, code.value


class StackLogic:
    """Stacking task logic with injected poses (orc_stack_logic)."""

    def __init__(self):
        self.L = lib()
        self.st = (C.c_int * 5)()
        self.reset()

    def reset(self):
        self.L.orc_stack_reset(self.st)

    def step(self, box21, target):
        box21, target = np.ascontiguousarray(box21, float).reshape(21), np.ascontiguousarray(target, float)
        obs = np.zeros(12, dtype=np.float32)
        succ, md = C.c_int(0), C.c_double(0)
        mode = C.create_string_buffer(4)
        self.L.orc_stack_logic(self.st, _p(box21), _p(target), _p(obs), C.byref(succ), C.byref(md), mode)
        return obs, bool(succ.value), md.value, mode.value.decode()
