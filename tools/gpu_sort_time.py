"""Sorting kernel timing on the GPU: per-step kernel time at rest on the platform and while every environment pushes a cube
(same script as tests/test_gpu_parity_sorting.py).  usage: python tools/gpu_sort_time.py [n_envs] [steps]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from d3il_amd.envs.sorting import SortingVecEnv, sample_contexts  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
iq = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ref_offline_ik.npz"))["sorting__traj_last"].copy()
env = SortingVecEnv(n, device=0)
env.set_init_qpos(iq)
t0 = time.time()
env.reset(context=sample_contexts(n, 4, seed=1))
torch.cuda.synchronize()
print("reset %.1f ms" % (1e3 * (time.time() - t0)), flush=True)
z = env.robot_state()[:, 2:3].clone()
des = env.obs[:, :2].to(torch.float64).clone()
quat = torch.tensor([0.0, 1, 0, 0], dtype=torch.float64, device=des.device).expand(n, 4)
for t in range(steps):
    box = env.obs[:, 2:4].to(torch.float64)
    if t < 12:
        target = des.clone()
    else:
        aligned = ((des[:, 0] - box[:, 0]).abs() < 0.008) & (des[:, 1] < box[:, 1] - 0.02)
        target = torch.where(aligned[:, None], torch.stack([box[:, 0], torch.full_like(box[:, 0], 0.36)], 1), box + torch.tensor([0.0, -0.06], dtype=torch.float64, device=box.device))
    d = target - des
    nn = d.norm(dim=1, keepdim=True)
    des = des + d / nn.clamp_min(1e-9) * torch.minimum(nn, torch.full_like(nn, 0.006))
    a = torch.cat([des, z, quat], dim=1).contiguous()
    torch.cuda.synchronize()
    t0 = time.time()
    env.step(a)
    torch.cuda.synchronize()
    dt = time.time() - t0
    fl = env.flags[:n].cpu().numpy()
    print("step %3d  %8.2f ms  fail %d overflow %d off %d  codes!=240: %d" % (t, 1e3 * dt, ((fl >> 16) & 1).sum(), ((fl >> 18) & 1).sum(), ((fl >> 19) & 1).sum(),
                                                                           int((env.mode != 240).sum())), flush=True)
