"""Stacking soak: workgroups of four identical copies under random joint motions with the gripper closing on nothing and opening again (random phase per
workgroup); counts environments whose state differs from position 0 of their workgroup and SOLVER_FAIL flags.  A position-dependent result is an
engine defect by construction.  usage (GPU box): [D3IL_LIB_PATH=...] python tools/gpu_stack_soak.py [envs] [steps] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from d3il_amd import capi  # noqa: E402
from d3il_amd.envs.stacking import CubeStackingVecEnv, load_test_contexts  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 5
wgs = n // 4
env = CubeStackingVecEnv(n, device=0)
q0, _, _ = env.start()
ctx = load_test_contexts()
wg = np.arange(n) // 4
env.reset(context=ctx[wg % 100])
rng = np.random.default_rng(seed)
cmd = np.tile(np.asarray(q0, dtype=np.float64), (wgs, 1))
phase, period = rng.integers(0, 40, size=wgs), rng.integers(20, 60, size=wgs)
bad_pos = np.zeros(4, dtype=np.int64)
first = None
dirty = np.zeros(wgs, dtype=bool)
fails = 0
for t in range(steps):
    cmd = cmd + rng.uniform(-0.01, 0.01, size=cmd.shape)
    grip = np.where((t + phase) % period < 0.7 * period, 0.0, 0.08)
    act = np.concatenate([cmd, grip[:, None]], axis=1)[wg]
    env.step(torch.as_tensor(act, dtype=torch.float64, device="cuda:0").contiguous())
    torch.cuda.synchronize()
    st, fl, _ = env.get_state()
    s4 = st.reshape(st.shape[0], wgs, 4)
    d = (s4 != s4[:, :, :1]).any(axis=0)                  # [wgs, 4]
    new = d.any(axis=1) & ~dirty
    if new.any():
        bad_pos += d[new].sum(axis=0)
        if first is None:
            first = (t, np.nonzero(new)[0][:6].tolist())
        dirty |= new
    fails = int(((fl & capi.FLAG_SOLVER_FAIL) != 0).sum())
print("lib %s: %d workgroups x 4 copies, %d steps (%.2e environment steps): workgroups whose copies separated %d (positions %s, first %s), environments with SOLVER_FAIL %d" % (
    os.path.basename(capi.lib_path()), wgs, steps, n * steps, int(dirty.sum()), bad_pos.tolist(), first, fails))
sys.exit(1 if dirty.any() or fails else 0)
