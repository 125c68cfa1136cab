"""Stacking kernel: clock ticks per phase of the physics sub-step (diagnostics build: D3IL_STATS_LIB=1), scripted pick-and-place.
usage: D3IL_STATS_LIB=1 python tools/gpu_stack_phases.py [n_envs]"""
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
names = ["arm dyn + tables", "boxes + static / box-box", "finger collision", "limits + aref", "solve", "integrate", "| warm gradient pass", "H0 / g0 / limits", "contact g+H pass", "cholesky", "tri solves", "Jp pass", "line search", "| collision: set-up", "box-box", "MPR"]
env.set_timing(True)
for lo, hi, label in ((0, 20, "rest / approach"), (60, 80, "grasp + lift"), (150, 170, "carry / place")):
    for t in range(lo if lo == 0 else 0, 0):
        pass
    buf0 = np.zeros(32 * 36 + 24); capi.check(env.L.d3il_debug_scratch(env.h, 0, buf0.ctypes.data_as(C.c_void_p), len(buf0)))
    ms = []
    for t in range(lo, hi):
        act = torch.as_tensor(np.stack([trajs[i % 4][min(t, len(trajs[i % 4]) - 1)] for i in range(min(n, 4))]), dtype=torch.float64, device=env.device)
        env.step(act[torch.arange(n, device=env.device) % 4].contiguous())
        torch.cuda.synchronize(); ms.append(env.last_step_ms())
    buf1 = np.zeros(32 * 36 + 24); capi.check(env.L.d3il_debug_scratch(env.h, 0, buf1.ctypes.data_as(C.c_void_p), len(buf1)))
    d = buf1[32 * 36 + 4:32 * 36 + 20] - buf0[32 * 36 + 4:32 * 36 + 20]
    print("%-16s steps %3d-%3d: kernel %.1f ms/step ; env 0 ticks/sub-step: %s  (total %.0f)" % (label, lo, hi, np.mean(ms), ", ".join("%s %.0f" % (nm, x / (30 * (hi - lo))) for nm, x in zip(names, d)), d.sum() / (30 * (hi - lo))))
