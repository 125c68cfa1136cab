import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from d3il_amd.envs.pushing import BlockPushVecEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
ctx60 = np.load(os.path.join(ROOT, "d3il_amd", "data", "pushing_test_contexts.npy"))
ctx = ctx60[np.arange(n) % 60]
iq = np.load(os.path.join(ROOT, "tests", "golden", "ref_offline_ik.npz"))["avoiding__traj_last"]
env = BlockPushVecEnv(n, device=0)
env.set_init_qpos(iq)
print("created", flush=True)
obs = env.reset(context=ctx)
torch.cuda.synchronize()
print("reset ok", obs[0].cpu().numpy(), flush=True)
des = env.robot_state()[:, :2].clone(); z = env.robot_state()[:, 2:3].clone()
quat = torch.tensor([0.0, 1, 0, 0], dtype=torch.float64, device=env.device).expand(n, 4)
for t in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
    act = torch.cat([des, z, quat], dim=1).contiguous()
    env.step(act)
    torch.cuda.synchronize()
    print("step", t, "ok", flush=True)
