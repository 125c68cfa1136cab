"""GPU smoke/debug run of the Pushing path against the oracle (run on the GPU box: python tools/gpu_push_debug.py [n] [steps])."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from d3il_amd.envs.pushing import BlockPushVecEnv  # noqa: E402
from d3il_amd.model import blob  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
ncheck = int(sys.argv[3]) if len(sys.argv) > 3 else 6
ctx60 = np.load(os.path.join(ROOT, "d3il_amd", "data", "pushing_test_contexts.npy"))
ctx = ctx60[np.arange(n) % 60]
iq = np.load(os.path.join(ROOT, "tests", "golden", "ref_offline_ik.npz"))["avoiding__traj_last"]
env = BlockPushVecEnv(n, device=0)
env.set_init_qpos(iq)
t0 = time.time()
obs = env.reset(context=ctx)
torch.cuda.synchronize()
print("reset %.3f s" % (time.time() - t0))
check = sorted(set(np.linspace(0, n - 1, ncheck).astype(int).tolist()))
oracles = []
for e in check:
    o = Oracle(blob.load("pushing"))
    o.env_start(iq)
    oo = o.push_reset(ctx[e])
    oracles.append(o)
    so, fo = o.push_state()
    st, fl, sc = env.get_state()
    print("env %d reset diff %.3e obs diff %.3e flags %x" % (e, np.max(np.abs(st[:68, e] - so)), np.max(np.abs(oo - obs[e].cpu().numpy())), fl[e]))
des = env.robot_state()[:, :2].clone()
z = env.robot_state()[:, 2:3].clone()
quat = torch.tensor([0.0, 1, 0, 0], dtype=torch.float64, device=env.device).expand(n, 4)
env.set_timing(True)
tms = []
for t in range(steps):
    o64 = env.obs.to(torch.float64)
    d = o64[:, 2:4] - des
    nn = d.norm(dim=1, keepdim=True).clamp_min(1e-9)
    des = des + d / nn * torch.minimum(nn, torch.full_like(nn, 0.006))
    act = torch.cat([des, z, quat], dim=1).contiguous()
    obs, rew, done, info = env.step(act)
    torch.cuda.synchronize()
    tms.append(env.last_step_ms())
    a = act.cpu().numpy()
    worst = 0
    for k, e in enumerate(check):
        oo, ro, do, io = oracles[k].push_step(a[e])
        so, fo = oracles[k].push_state()
        st, fl, sc = env.get_state()
        dd = np.abs(st[:68, e] - so)
        pos_idx = list(range(0, 9)) + list(range(25, 28)) + list(range(42, 49)) + list(range(55, 62))
        worst = max(worst, dd[pos_idx].max())
        if t % 5 == 0 or dd[pos_idx].max() > 1e-6:
            print("t %d env %d posdiff %.2e alldiff %.2e (%d) ncon %d flags %x mode %d/%d done %d/%d md %.2e" % (
                t, e, dd[pos_idx].max(), dd.max(), int(dd.argmax()), fo[6], fl[e], int(info["mode"][e]), io["mode"], int(done[e]), do,
                abs(float(info["mean_distance"][e]) - io["mean_distance"])))
    if worst > 1e-3:
        print("DIVERGED")
        break
print("kernel ms: first %.3f mean %.3f min %.3f max %.3f" % (tms[0], np.mean(tms), np.min(tms), np.max(tms)))
st, fl, sc = env.get_state()
print("flag summary: solver_fail %d overflow %d off_table %d" % (np.sum((fl >> 16) & 1), np.sum((fl >> 18) & 1), np.sum((fl >> 19) & 1)))
print("env-steps/s at this n: %.0f" % (n / (np.mean(tms) * 1e-3)))
