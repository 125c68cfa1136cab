"""Determinism probe: copies of one context placed in different lanes / workgroups must evolve bit-identically."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from d3il_amd.envs.sorting import SortingVecEnv, sample_contexts
from tests.dist_sim_worker import PushToBinAgent
nctx, rep = 10, 7
ctx = np.repeat(sample_contexts(nctx, 4, seed=0), rep, axis=0)
n = nctx * rep
env = SortingVecEnv(n, device=0); env.start(); obs = env.reset(context=ctx)
ag = PushToBinAgent()
des = env.robot_state()[:, :2].clone(); z = env.robot_state()[:, 2:3].clone()
quat = torch.tensor([0.0, 1, 0, 0], dtype=torch.float64, device=env.device).expand(n, 4)
for t in range(140):
    oin = torch.cat((des, obs.to(torch.float64)), 1)
    des = des + ag.predict_batch(oin)
    obs, _, done, info = env.step(torch.cat((des, z, quat), 1).contiguous())
    torch.cuda.synchronize()
    st, fl, sc = env.get_state()
    s = st[:127].reshape(127, nctx, rep)
    dev = np.abs(s - s[:, :, :1]).max(axis=(0, 2))
    if dev.max() > 0:
        c = int(np.argmax(dev)); rows = np.nonzero(np.abs(s[:, c, :] - s[:, c, :1]).max(axis=1))[0]
        print("t", t, "first deviation: ctx", c, "max", dev.max(), "rows", rows[:20]); break
else:
    print("bit-identical over 140 steps")
