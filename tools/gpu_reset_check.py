import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from d3il_amd.envs.avoiding import ObstacleAvoidanceVecEnv
g = np.load("tests/golden/oracle_avoiding_rollout.npz")
env = ObstacleAvoidanceVecEnv(128, device=0)
env.set_init_qpos(g["init_qpos"])
env.reset(); torch.cuda.synchronize()
st, fl, sc = env.get_state()
d = st[:, 0] - g["random__states"][0]
print("reset diff per field:", np.array2string(d, precision=3, max_line_width=200))
print("flags", hex(fl[0]), "all lanes equal", (st == st[:, :1]).all())
a = torch.as_tensor(np.tile(g["random__actions"][0], (128, 1)), dtype=torch.float64, device=env.device).contiguous()
env.step(a); torch.cuda.synchronize()
st, fl, sc = env.get_state()
d = st[:, 0] - g["random__states"][1]
print("step1 diff per field:", np.array2string(d, precision=3, max_line_width=200))
