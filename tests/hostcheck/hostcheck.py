"""ctypes loader for the TEST-ONLY host build of the kernel math (tests/hostcheck/hostcheck.cpp)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "libd3il_hostcheck.so")
    srcs = [os.path.join(_HERE, "hostcheck.cpp")] + [os.path.join(_HERE, "..", "..", "d3il_amd", "csrc", f) for f in ("panda_step.h", "panda_consts.h", "rigid_common.h", "gen_step.h", "gen_tree.h", "stack_step.h")]
    srcs.append(os.path.join(_HERE, "..", "..", "include", "d3il_model_blob.h"))
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", so, srcs[0]])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.hc_create.restype = C.c_void_p
        _LIB.hc_sizeof_consts.restype = C.c_int
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class HostCheck:
    def __init__(self, blob):
        self.L = lib()
        err = C.c_char_p()
        self.h = C.c_void_p(self.L.hc_create(C.byref(blob), C.byref(err)))
        if not self.h:
            raise RuntimeError("hc_create: %s" % (err.value.decode() if err.value else "?"))

    def consts(self):
        a, b, m, c = np.zeros(9), np.zeros(1), np.zeros(7), np.zeros((7, 3))
        self.L.hc_get_consts(self.h, _p(a), _p(b), _p(m), _p(c))
        return a, b[0], m, c

    def dynamics(self, q, v):
        q, v = np.ascontiguousarray(q, float), np.ascontiguousarray(v, float)
        M, b, t = np.zeros((9, 9)), np.zeros(9), np.zeros(3)
        self.L.hc_dynamics(self.h, _p(q), _p(v), _p(M), _p(b), _p(t))
        return M, b, t

    def ik_fk(self, q):
        q = np.ascontiguousarray(q, float)
        pos, quat, J = np.zeros(3), np.zeros(4), np.zeros((6, 7))
        self.L.hc_ik_fk(self.h, _p(q), _p(pos), _p(quat), _p(J))
        return pos, quat, J

    def ik_control(self, setpoint, cur_q, cur_v, ikq, ikqd, flags, fast):
        sp, cq, cv = (np.ascontiguousarray(x, float) for x in (setpoint, cur_q, cur_v))
        f = C.c_int(flags)
        tau = np.zeros(7)
        self.L.hc_ik_control(self.h, _p(sp), _p(cq), _p(cv), _p(ikq), _p(ikqd), C.byref(f), int(fast), _p(tau))
        return tau, f.value

    def physics_substep(self, s, f, tau, ff):
        tau, ff = np.ascontiguousarray(tau, float), np.ascontiguousarray(ff, float)
        self.L.hc_physics_substep(self.h, _p(s), _p(f), _p(tau), _p(ff))

    def env_reset(self, init_qpos):
        init_qpos = np.ascontiguousarray(init_qpos, float)
        s, f, obs = np.zeros(42), np.zeros(2, dtype=np.int32), np.zeros(2, dtype=np.float32)
        self.L.hc_env_reset(self.h, _p(init_qpos), _p(s), _p(f), _p(obs))
        return s, f, obs

    def env_step(self, s, f, action, fast=True):
        action = np.ascontiguousarray(action, float)
        obs, done = np.zeros(2, dtype=np.float32), np.zeros(1, dtype=np.uint8)
        self.L.hc_env_step(self.h, _p(s), _p(f), _p(action), _p(obs), _p(done), int(fast))
        return obs, bool(done[0])


class GenHostCheck:
    """Host build of the generic engine (d3il_amd/csrc/gen_step.h) running the Sorting task, one environment."""

    def __init__(self, blob):
        self.L = lib()
        self.L.hc_gen_create.restype = C.c_void_p
        err = C.c_char_p()
        self.h = C.c_void_p(self.L.hc_gen_create(C.byref(blob), C.byref(err)))
        if not self.h:
            raise RuntimeError("hc_gen_create: %s" % (err.value.decode() if err.value else "?"))
        nb, ns = C.c_int(0), C.c_int(0)
        st = np.zeros((32, 7))
        self.n = self.L.hc_gen_info(self.h, C.byref(nb), C.byref(ns), _p(st))
        self.nb, self.ns, self.statics = nb.value, ns.value, st[: ns.value].copy()
        ws = np.zeros(8)
        self.ns_core = self.L.hc_gen_workspace(self.h, _p(ws))      # statics [0, ns_core) stand inside the table, the rest are the frame beams around it
        self.inner, self.outer = ws[:4].copy(), ws[4:].copy()      # (lo x, lo y, hi x, hi y) of the beam-free region / of the modelled workspace
        self.n_obs = 2 + 3 * self.nb
        self.s = np.zeros(self.n)
        self.f = np.zeros(2, dtype=np.int32)

    def reset(self, init_qpos, ctx):
        init_qpos, ctx = np.ascontiguousarray(init_qpos, float), np.ascontiguousarray(ctx, float).reshape(7 * self.nb)
        obs = np.zeros(20, dtype=np.float32)
        self.L.hc_gen_reset(self.h, _p(init_qpos), _p(ctx), _p(self.s), _p(self.f), _p(obs))
        return obs[: self.n_obs]

    def step(self, action, fast=True):
        action = np.ascontiguousarray(action, float)
        obs = np.zeros(20, dtype=np.float32)
        done, code = C.c_ubyte(0), C.c_int(0)
        self.L.hc_gen_step(self.h, _p(self.s), _p(self.f), _p(action), _p(obs), C.byref(done), C.byref(code), int(fast))
        fl = int(self.f[0]) & 0xFFFFFFFF
        return obs[: self.n_obs], bool(done.value), dict(mode=code.value, success=bool(fl & (1 << 13)), flags=fl)

    def substep(self, tau, ff):
        tau, ff = np.ascontiguousarray(tau, float), np.ascontiguousarray(ff, float)
        self.L.hc_gen_substep(self.h, _p(self.s), _p(self.f), _p(tau), _p(ff))

    def box(self, b):
        o = 42 + 13 * b
        return self.s[o:o + 3], self.s[o + 3:o + 7], self.s[o + 7:o + 13]


class PushHostCheck(GenHostCheck):
    """The Pushing task on the host build of the generic engine (gen_step.h, GEN_TASK_PUSHING - the engine the product runs the task on; the round-1 Pushing
    engine is gone), one environment, with the return values of Block_Push_Env.step: state rows 0 .. 88 = arm, cubes, warm start; row 89 info['mean_distance'],
    row 90 the reward; first-visit / mode bits in the flag word."""

    def __init__(self, blob):
        super().__init__(blob)
        assert (self.nb, self.n_obs, self.n) == (2, 8, 91)

    def step(self, action, fast=True):
        obs, done, info = super().step(action, fast)
        fl = info["flags"]
        mode = info["mode"] if info["mode"] < 32768 else info["mode"] - 65536
        return obs, float(self.s[90]), done, dict(mode=mode, success=info["success"], mean_distance=float(self.s[89]), first_visit=(fl & 7) - 1, flags=fl)


class StackHostCheck:
    """Host build of the Stacking engine (d3il_amd/csrc/stack_step.h), one environment."""

    def __init__(self, blob):
        self.L = lib()
        self.L.hc_stack_create.restype = C.c_void_p
        err = C.c_char_p()
        self.h = C.c_void_p(self.L.hc_stack_create(C.byref(blob), C.byref(err)))
        if not self.h:
            raise RuntimeError("hc_stack_create: %s" % (err.value.decode() if err.value else "?"))
        self.n = self.L.hc_stack_state_size()
        self.s = np.zeros(self.n)
        self.f = np.zeros(2, dtype=np.int32)

    def reset(self, init_qpos, ctx):
        init_qpos, ctx = np.ascontiguousarray(init_qpos, float), np.ascontiguousarray(ctx, float).reshape(21)
        obs = np.zeros(12, dtype=np.float32)
        self.L.hc_stack_reset(self.h, _p(init_qpos), _p(ctx), _p(self.s), _p(self.f), _p(obs))
        return obs

    def step(self, action8):
        action8 = np.ascontiguousarray(action8, float).reshape(8)
        obs = np.zeros(12, dtype=np.float32)
        done, md = C.c_ubyte(0), C.c_double(0)
        self.L.hc_stack_step(self.h, _p(self.s), _p(self.f), _p(action8), _p(obs), C.byref(done), C.byref(md))
        fl = int(self.f[0]) & 0xFFFFFFFF
        n = fl & 3
        mode = "".join("rgb"[(fl >> (2 + 2 * i)) & 3] for i in range(n))
        return obs, bool(done.value), dict(mode=mode, success=bool(fl & (1 << 13)), mean_distance=md.value, flags=fl)

    def scratch(self, count):
        """First ``count`` doubles of the solver scratch area (contact records 32 x 36, then the diagnostics of the last sub-step)."""
        out = np.zeros(count)
        self.L.hc_stack_scratch(self.h, _p(out), count)
        return out

    def contacts(self):
        ncon = int(self.scratch(32 * 36 + 4)[32 * 36 + 3])
        out = np.zeros((48, 10))
        n = self.L.hc_stack_contacts(self.h, _p(out))
        return out[:min(n, ncon)]
