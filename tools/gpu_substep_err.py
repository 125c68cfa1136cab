"""Where does the one-step difference between the HIP path and the oracle come from?  Pushing: identical mid-episode states, then
k = 1, 2, 5, 35 physics sub-steps on both sides (environments built with n_substeps = k); production and strict solver rules.
Also times the contact regime in both modes (shows that the strict rule is live).  Run on the GPU box."""
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from d3il_amd.envs.pushing import BlockPushVecEnv  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

G = np.load(os.path.join(ROOT, "tests", "golden", "ref_offline_ik.npz"))
iq = G["avoiding__traj_last"].copy()
ctx60 = np.load(os.path.join(ROOT, "d3il_amd", "data", "pushing_test_contexts.npy"))
P_POS = list(range(0, 9)) + list(range(42, 49)) + list(range(55, 62))
P_VEL = list(range(9, 18)) + list(range(49, 55)) + list(range(62, 68))
n = 120


def action(des, z):
    quat = torch.tensor([0.0, 1, 0, 0], dtype=torch.float64, device=des.device).expand(des.shape[0], 4)
    return torch.cat([des, z, quat], dim=1).contiguous()


# 1. a rollout that produces mid-episode states (35 sub-steps), sampled at a few instants
env = BlockPushVecEnv(n, device=0)
env.set_init_qpos(iq)
env.reset(context=ctx60[np.arange(n) % 60])
des = env.robot_state()[:, :2].clone(); z = env.robot_state()[:, 2:3].clone()
samples = []
env.set_timing(True)
for t in range(70):
    o64 = env.obs.to(torch.float64)
    d = (o64[:, 2:4] if t < 45 else o64[:, 5:7]) - des
    nn = d.norm(dim=1, keepdim=True).clamp_min(1e-9)
    des = des + d / nn * torch.minimum(nn, torch.full_like(nn, 0.006))
    act = action(des, z)
    if t in (30, 40, 50, 60, 68):
        torch.cuda.synchronize()
        samples.append((env.get_state(), act.cpu().numpy().copy(), act.clone()))
    env.step(act)
env.close()

for strict in (0, 1):
    for k in (1, 2, 5, 35):
        e2 = BlockPushVecEnv(n, device=0, n_substeps=k)
        e2.set_option("solver_strict", strict)
        e2.set_init_qpos(iq)
        e2.reset(context=ctx60[np.arange(n) % 60])
        o = Oracle(e2.blob); o.env_start(iq)
        wp = wv = 0.0
        ncon = []
        for (st0, fl0, sc0), a, act in samples:
            e2.set_state(st0, fl0, sc0)
            e2.step(act)
            torch.cuda.synchronize()
            st1, fl1, sc1 = e2.get_state()
            for e in range(0, n, 5):
                o.push_set_state(st0[:68, e], step=sc0[e], terminated=bool(fl0[e] & (1 << 12)), first_visit=int(fl0[e] & 7) - 1, ik_valid=bool(fl0[e] & (1 << 15)))
                o.push_step(a[e])
                so, _ = o.push_state()
                ep, ev = np.abs(st1[P_POS, e] - so[P_POS]).max(), np.abs(st1[P_VEL, e] - so[P_VEL]).max()
                wp, wv = max(wp, ep), max(wv, ev)
                ncon.append(len(o.contacts()))
        print("strict=%d  sub-steps=%2d : max |dpos| %.3e  max |dvel| %.3e   (oracle contacts: min %d max %d)" % (strict, k, wp, wv, min(ncon), max(ncon)), flush=True)
        e2.close()

# 2. is the strict rule live?  kernel time of contact steps in both modes
for strict in (0, 1):
    env = BlockPushVecEnv(1024, device=0)
    env.set_option("solver_strict", strict)
    env.set_init_qpos(iq)
    env.reset(context=ctx60[np.arange(1024) % 60])
    des = env.robot_state()[:, :2].clone(); z = env.robot_state()[:, 2:3].clone()
    env.set_timing(True)
    ms = []
    for t in range(60):
        o64 = env.obs.to(torch.float64)
        d = o64[:, 2:4] - des
        nn = d.norm(dim=1, keepdim=True).clamp_min(1e-9)
        des = des + d / nn * torch.minimum(nn, torch.full_like(nn, 0.006))
        env.step(action(des, z))
        torch.cuda.synchronize()
        ms.append(env.last_step_ms())
    print("strict=%d  contact steps 35-60: %.3f ms per step" % (strict, float(np.mean(ms[35:]))), flush=True)
    env.close()
