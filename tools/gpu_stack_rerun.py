"""Stacking diagnostics: along a scripted rollout at sub-step resolution, every device step is executed twice from the same state
(get_state -> step -> set_state(saved) -> step): a difference means the result depends on something outside (state, flags, step)."""
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from d3il_amd.controllers.scripted_stacking import build_trajectory  # noqa: E402
from d3il_amd.envs.stacking import CubeStackingVecEnv  # noqa: E402
from d3il_amd.model import blob as blob_mod  # noqa: E402

js = blob_mod.load_json("stacking")
ctx100 = np.load(os.path.join(ROOT, "d3il_amd", "data", "stacking_test_contexts.npy"))
ids = [1, 5, 17, 60, 77, 93]
n = len(ids)
env = CubeStackingVecEnv(n, device=0, n_substeps=1, max_steps_per_episode=1000000)
q0, _, _ = env.start()
env.reset(context=ctx100[ids])
trajs = [build_trajectory(js, q0, ctx100[i], n_boxes=2, speed=1.0) for i in ids]
nd = 0
for tt in range(130 * 30):
    t = tt // 30
    act = torch.as_tensor(np.stack([trajs[k][t] for k in range(n)]), dtype=torch.float64, device=env.device).contiguous()
    if not (45 <= t < 56 or 116 <= t < 124):
        env.step(act)
        continue
    torch.cuda.synchronize()
    st0, fl0, sc0 = env.get_state()
    env.step(act); torch.cuda.synchronize()
    sa, fa, ca = env.get_state()
    env.set_state(st0, fl0, sc0)
    env.step(act); torch.cuda.synchronize()
    sb, fb, cb = env.get_state()
    dd = np.abs(sa - sb).max(axis=0)
    if dd.max() > 0:
        nd += 1
        if nd <= 12:
            print("sub-step %d (t %d): first run != second run, per env max diff %s" % (tt, t, np.array2string(dd, precision=2)), flush=True)
    env.set_state(sa, fa, ca)
print("sub-steps with a difference:", nd)
