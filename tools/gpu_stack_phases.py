"""Stacking kernel: clock ticks (100 MHz) per phase of the physics sub-step (diagnostics build: D3IL_STATS_LIB=1), scripted pick-and-place.
usage: D3IL_STATS_LIB=1 python tools/gpu_stack_phases.py [n_envs]
Per-environment phases (slots 0, 3, 5) and the phases of the cooperative solve of environment 0 (7 .. 12) are read from the diagnostics words
of environment 0; the solver and collision phases are timed by lane 0 of workgroup 0."""
import ctypes as C
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from d3il_amd import capi  # noqa: E402
from d3il_amd.controllers.scripted_stacking import build_trajectory  # noqa: E402
from d3il_amd.envs.stacking import CubeStackingVecEnv, load_test_contexts  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = CubeStackingVecEnv(n, device=0)
q0, _, _ = env.start()
ctx = load_test_contexts()[:4]
env.reset(context=ctx[np.arange(n) % 4])
trajs = [build_trajectory(env.js, q0, c, speed=0.7) for c in ctx]
names = ["arm dyn + tables", "-", "-", "build rows", "-", "integrate", "-", "-", "contact g+H pass", "cholesky", "tri solves", "Jp pass", "line search"]
env.set_timing(True)
nwg = (n + 3) // 4


def snap():
    buf = np.zeros(32 * 36 + 24)
    capi.check(env.L.d3il_debug_scratch(env.h, 0, buf.ctypes.data_as(C.c_void_p), len(buf)))
    st = (C.c_uint64 * 32)()
    capi.check(env.L.d3il_debug_stats(st, 0))
    return buf[32 * 36 + 4:32 * 36 + 17].copy(), np.array([st[16 + k] for k in (13, 14, 15, 3, 8, 9, 10, 11, 12, 6, 7, 0, 1, 2, 4, 5)] + [st[k] for k in (8, 9, 10, 11, 12)], dtype=float)


t = 0
windows = ((0, 20, "rest / approach"), (60, 80, "grasp + lift"), (150, 170, "carry / place"))
if len(sys.argv) > 2 and sys.argv[2] == "scan":      # the whole scripted episode in windows of 10 steps every 50: where is the slowest phase?
    T = max(len(tr) for tr in trajs)
    windows = tuple((k, k + 10, "steps %d.." % k) for k in range(0, T, 50))
for lo, hi, label in windows:
    while t < lo:
        act = torch.as_tensor(np.stack([trajs[i % 4][min(t, len(trajs[i % 4]) - 1)] for i in range(4)]), dtype=torch.float64, device=env.device)
        env.step(act[torch.arange(n, device=env.device) % 4].contiguous()); t += 1
    torch.cuda.synchronize()
    a0, c0 = snap()
    ms = []
    while t < hi:
        act = torch.as_tensor(np.stack([trajs[i % 4][min(t, len(trajs[i % 4]) - 1)] for i in range(4)]), dtype=torch.float64, device=env.device)
        env.step(act[torch.arange(n, device=env.device) % 4].contiguous()); t += 1
        torch.cuda.synchronize(); ms.append(env.last_step_ms())
    a1, c1 = snap()
    d, dc = (a1 - a0) / (30 * (hi - lo)), (c1 - c0) / (30 * (hi - lo))
    print("%-16s steps %3d-%3d: kernel %.2f ms/step ; ticks per sub-step, environment 0: %s | collision (per workgroup): set-up %.0f, box-box %.0f, MPR %.0f" % (
        label, lo, hi, np.mean(ms), ", ".join("%s %.0f" % (nm, x) for nm, x in zip(names, d) if nm != "-"), dc[0], dc[1], dc[2]))
    print("    dual solves per sub-step and workgroup %.2f (Newton iterations %.2f): build rows %.0f, g+H pass %.0f, cholesky %.0f, tri solves %.0f, Jp %.0f, line search %.0f ticks per sub-step and workgroup" % (
        dc[9], dc[10], dc[3], dc[4], dc[5], dc[6], dc[7], dc[8]))
    print("    start point (twists of x, J x - aref) %.0f; per pass: contact lanes (cone, K, F, LDS adds) %.0f, aggregation %.0f, gradient %.0f; Hessian rows (passes that go on) %.0f; max |g| is 'g+H pass'" % (dc[11], dc[12], dc[13], dc[14], dc[15]))
    print("    MPR phase of workgroup 0 per sub-step: finger-hull batches %.2f (%.0f ticks incl. the job set-up), hand jobs %.2f with %.1f hand support calls (%.0f ticks)" % (dc[16], dc[19], dc[17], dc[18], dc[20]))
    print("    kernel time per sub-step: %.0f ticks" % (np.mean(ms) * 1e-3 / 30 * 1e8))
