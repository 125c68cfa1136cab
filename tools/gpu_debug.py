import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from d3il_amd.envs.avoiding import ObstacleAvoidanceVecEnv
g = np.load("tests/golden/oracle_avoiding_rollout.npz")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
bad_total = 0
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 2):
  for fast in (1, 0):
    for name in ("random", "collide", "succeed", "zigzag"):
        env = ObstacleAvoidanceVecEnv(n, device=0)
        env.set_option("ik_fast_path", fast)
        env.set_init_qpos(g["init_qpos"])
        env.reset(); torch.cuda.synchronize()
        acts = g[name + "__actions"]
        first = None
        for t in range(len(acts)):
            a = torch.as_tensor(np.tile(acts[t], (n, 1)), dtype=torch.float64, device=env.device).contiguous()
            env.step(a); torch.cuda.synchronize()
            st, fl, sc = env.get_state()
            err = np.abs(st - g[name + "__states"][t + 1][:, None]).max(0)
            if (err > 1e-8).any() and first is None:
                first = (t, np.where(err > 1e-8)[0], err.max())
                break
        print(rep, "fast", fast, name, "steps", len(acts), "OK" if first is None else "FAIL at t=%d lanes=%s maxerr=%g" % (first[0], first[1][:24], first[2]))
        bad_total += first is not None
        env.close()
print("failures", bad_total)
