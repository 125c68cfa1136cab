"""Stacking: Newton iterations of the cooperative solver in the LAST sub-step of a step, over environments and phases (diagnostics)."""
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

n = 64
env = CubeStackingVecEnv(n, device=0)
q0, _, _ = env.start()
ctx = load_test_contexts()[:4]
env.reset(context=ctx[np.arange(n) % 4])
trajs = [build_trajectory(env.js, q0, c, speed=0.5) for c in ctx]
T = min(len(t) for t in trajs)
hist = {}
for t in range(0, 400):
    act = torch.as_tensor(np.stack([trajs[i % 4][t] for i in range(n)]), dtype=torch.float64, device=env.device)
    env.step(act.contiguous())
    if t % 10 == 9:
        its, ncs = [], []
        for e in range(4):
            buf = np.zeros(32 * 36 + 4); capi.check(env.L.d3il_debug_scratch(env.h, e, buf.ctypes.data_as(C.c_void_p), len(buf)))
            its.append(buf[32 * 36]); ncs.append(buf[32 * 36 + 3])
        print("step %3d  iterations (last sub-step, env 0..3): %s   contacts: %s" % (t, its, ncs))
