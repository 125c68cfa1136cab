"""Pushing / Sorting: results must not depend on where an environment sits in the batch.  Every environment has its own sampled context and is driven by the
closed-loop scripted policy (which reads only that environment's observation: rod pushes, cube <-> cube and cube <-> wall contacts); the batch runs twice,
the second time PERMUTED (other workgroup, other lane, other workgroup mates).  Each environment's state must be bit-identical in both runs at every
step.  The Stacking counterpart is tools/gpu_stack_perm.py.
usage (GPU box): python tools/gpu_perm_push_sort.py [pushing|sorting] [envs] [steps] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from d3il_amd import capi  # noqa: E402
from d3il_amd.agents import ScriptedGoalPushPolicy  # noqa: E402

task = sys.argv[1] if len(sys.argv) > 1 else "pushing"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 250
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dev = torch.device("cuda:0")
rng = np.random.default_rng(seed)
if task == "pushing":
    from d3il_amd.envs.pushing import BlockPushVecEnv as Env, sample_contexts
    ctx = sample_contexts(n, seed=seed)
    plan = rng.integers(0, 4, size=n)
else:
    from d3il_amd.envs.sorting import SortingVecEnv as Env, sample_contexts
    ctx = sample_contexts(n, 4, seed=seed).reshape(n, -1)
    plan = None
perm = rng.permutation(n)


def run(order):
    env = Env(n, device=0)
    env.start()
    obs = env.reset(random=False, context=ctx[order])
    pol = ScriptedGoalPushPolicy(task, plan=None if plan is None else plan[order], device=dev)
    rs = env.robot_state()
    des, z = rs[:, :2].clone(), rs[:, 2:3].clone()
    quat = torch.tensor([0.0, 1.0, 0.0, 0.0], dtype=torch.float64, device=dev).expand(n, 4)
    inv = np.argsort(order)
    out, flags = [], None
    for t in range(steps):
        des = des + pol.predict_batch(torch.cat((des, obs.to(torch.float64)), dim=1))
        obs, _, done, info = env.step(torch.cat((des, z, quat), dim=1).contiguous())
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
moved = int((np.abs(a[-1][42:44] - a[0][42:44]).max(axis=0) > 1e-3).sum())
print("%s, lib %s: %d environments, %d steps (first cube moved in %d of them): environments whose two runs differ %d (first %s); SOLVER_FAIL run 1 %d, run 2 %d" % (
    task, os.path.basename(capi.lib_path()), n, steps, moved, int(dirty.sum()), first,
    int(((fa & capi.FLAG_SOLVER_FAIL) != 0).sum()), int(((fb & capi.FLAG_SOLVER_FAIL) != 0).sum())))
sys.exit(1 if dirty.any() else 0)
