import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from d3il_amd.envs.avoiding import ObstacleAvoidanceVecEnv
g = np.load("tests/golden/oracle_avoiding_rollout.npz")
n = 128
env = ObstacleAvoidanceVecEnv(n, device=0)
env.set_init_qpos(g["init_qpos"])
env.reset(); torch.cuda.synchronize()
st, fl, sc = env.get_state()
print("reset: identical lanes", (st == st[:, :1]).all(), "finite", np.isfinite(st).all())
acts = g["random__actions"]
for t in range(len(acts)):
    a = torch.as_tensor(np.tile(acts[t], (n, 1)), dtype=torch.float64, device=env.device).contiguous()
    env.step(a); torch.cuda.synchronize()
    st, fl, sc = env.get_state()
    same = (st == st[:, :1])
    if not same.all() or not np.isfinite(st).all():
        bad_env = np.where(~same.all(0))[0]
        bad_field = np.where(~same.all(1))[0]
        print("t", t, "bad envs", bad_env[:20], len(bad_env), "fields", bad_field, "maxdiff", np.nanmax(np.abs(st - st[:, :1])), "finite", np.isfinite(st).all())
        print("flags", [hex(x) for x in np.unique(fl)])
        e = bad_env[0]
        print("env", e, st[:, e] - st[:, 0])
        print("golden diff env0", np.abs(st[:, 0] - g["random__states"][t + 1]).max(), "env bad", np.abs(st[:, e] - g["random__states"][t + 1]).max())
        break
else:
    print("all steps identical")
