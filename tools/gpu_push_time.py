"""Kernel timing of the Pushing step in two regimes (run on the GPU box): cubes at rest, and rod pushing the red cube."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from d3il_amd.envs.pushing import BlockPushVecEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ctx60 = np.load(os.path.join(ROOT, "d3il_amd", "data", "pushing_test_contexts.npy"))
ctx = ctx60[np.arange(n) % 60]
iq = np.load(os.path.join(ROOT, "tests", "golden", "ref_offline_ik.npz"))["avoiding__traj_last"]
env = BlockPushVecEnv(n, device=0)
env.set_init_qpos(iq)
env.reset(context=ctx)
des = env.robot_state()[:, :2].clone(); z = env.robot_state()[:, 2:3].clone()
quat = torch.tensor([0.0, 1, 0, 0], dtype=torch.float64, device=env.device).expand(n, 4)
env.set_timing(True)
ms = []
for t in range(70):
    if t >= 12:
        o64 = env.obs.to(torch.float64)
        d = o64[:, 2:4] - des
        nn = d.norm(dim=1, keepdim=True).clamp_min(1e-9)
        des = des + d / nn * torch.minimum(nn, torch.full_like(nn, 0.006))
    env.step(torch.cat([des, z, quat], dim=1).contiguous())
    torch.cuda.synchronize()
    ms.append(env.last_step_ms())
ms = np.array(ms)
print("settling steps 0-3 :", np.round(ms[:4], 3))
print("rest steps 6-11    : mean %.3f" % ms[6:12].mean())
print("approach 12-25     : mean %.3f" % ms[12:26].mean())
print("contact 35-70      : mean %.3f min %.3f max %.3f" % (ms[35:].mean(), ms[35:].min(), ms[35:].max()))
st, fl, sc = env.get_state()
print("flags: fail %d overflow %d offtable %d" % (np.sum((fl >> 16) & 1), np.sum((fl >> 18) & 1), np.sum((fl >> 19) & 1)))
