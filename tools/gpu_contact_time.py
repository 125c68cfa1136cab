"""Avoiding: kernel time with every environment's rod pressed against the first obstacle (steady contact regime)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from d3il_amd.envs.avoiding import ObstacleAvoidanceVecEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = ObstacleAvoidanceVecEnv(n, device=0, max_steps_per_episode=100000)
env.start(); env.reset()
tcp = env.robot_state().clone()
des = tcp[:, :2].clone(); z = tcp[:, 2:3].clone()
quat = torch.tensor([0.0, 1, 0, 0], dtype=torch.float64, device=env.device).expand(n, 4)
target = torch.tensor([0.5, -0.1], dtype=torch.float64, device=env.device)
env.set_timing(True)
ms, nc = [], []
for t in range(80):
    d = target - des
    nn = d.norm(dim=1, keepdim=True).clamp_min(1e-9)
    des = des + d / nn * torch.minimum(nn, torch.full_like(nn, 0.005))
    env.step(torch.cat([des, z, quat], dim=1).contiguous())
    torch.cuda.synchronize()
    ms.append(env.last_step_ms())
    nc.append(int(((env.flags[:n] >> 14) & 1).sum()))
ms = np.array(ms)
for i in range(0, 80, 10):
    print(i, np.round(ms[i:i + 10], 3), nc[i + 9])
