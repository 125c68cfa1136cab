"""Stacking soak 2: every environment has its own context and its own open-loop action sequence (random joint motions, gripper closing on nothing and
opening again with a random phase); the batch is run twice, the second time with the environments PERMUTED (other workgroup mates, other workgroup
positions).  Each environment's state must be bit-identical in both runs at every step: the engine's results may not depend on where an environment
sits.  usage (GPU box): [D3IL_LIB_PATH=...] python tools/gpu_stack_perm.py [envs] [steps] [seed]"""
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
ctx = load_test_contexts()
rng = np.random.default_rng(seed)
cid = rng.integers(0, 100, size=n)
phase, period = rng.integers(0, 40, size=n), rng.integers(20, 60, size=n)
delta = rng.uniform(-0.01, 0.01, size=(steps, n, 7))
perm = rng.permutation(n)


def run(order):
    """order[k] = the logical environment that sits in slot k; returns the per-step states in LOGICAL order."""
    env = CubeStackingVecEnv(n, device=0)
    q0, _, _ = env.start()
    env.reset(context=ctx[cid[order]])
    cmd = np.tile(np.asarray(q0, dtype=np.float64), (n, 1))
    inv = np.argsort(order)
    out, flags = [], None
    for t in range(steps):
        cmd = cmd + delta[t]
        grip = np.where((t + phase) % period < 0.7 * period, 0.0, 0.08)
        act = np.concatenate([cmd, grip[:, None]], axis=1)[order]
        env.step(torch.as_tensor(act, dtype=torch.float64, device="cuda:0").contiguous())
        torch.cuda.synchronize()
        st, fl, _ = env.get_state()
        out.append(st[:, inv].copy()); flags = fl[inv].copy()
    env.close()
    return out, flags


a, fa = run(np.arange(n))
b, fb = run(perm)
dirty = np.zeros(n, dtype=bool)
first = None
for t in range(steps):
    d = (a[t] != b[t]).any(axis=0) & ~dirty
    if d.any() and first is None:
        first = (t, np.nonzero(d)[0][:6].tolist())
    dirty |= d
slot_a, slot_b = np.arange(n) % 4, np.argsort(perm) % 4
print("lib %s: %d environments, %d steps: environments whose two runs differ %d (first %s; workgroup position in run 1 %s, in run 2 %s); SOLVER_FAIL run 1 %d, run 2 %d" % (
    os.path.basename(capi.lib_path()), n, steps, int(dirty.sum()), first, np.bincount(slot_a[dirty], minlength=4).tolist(), np.bincount(slot_b[dirty], minlength=4).tolist(),
    int(((fa & capi.FLAG_SOLVER_FAIL) != 0).sum()), int(((fb & capi.FLAG_SOLVER_FAIL) != 0).sum())))
sys.exit(1 if dirty.any() else 0)
