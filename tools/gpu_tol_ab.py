"""Parity of the HIP path against the CPU oracle as a function of the contact solvers' stopping rule (run on the GPU box).

For Pushing and Sorting, with solver_strict = 0 (production) and 1 (the oracle's rule):
  * one-step error from identical mid-episode states (positions / velocities separately, max over the sampled envs);
  * the horizon (env steps) over which followed environments stay within 1e-8 / 1e-6 / 1e-4 (the north star) of the oracle.
usage: python tools/gpu_tol_ab.py [pushing|sorting|both]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from d3il_amd.model import blob as blob_mod  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

G = np.load(os.path.join(ROOT, "tests", "golden", "ref_offline_ik.npz"))


def action(des, z):
    n = des.shape[0]
    quat = torch.tensor([0.0, 1, 0, 0], dtype=torch.float64, device=des.device).expand(n, 4)
    return torch.cat([des, z, quat], dim=1).contiguous()


P_POS = list(range(0, 9)) + list(range(25, 28)) + list(range(42, 49)) + list(range(55, 62))
P_VEL = list(range(9, 18)) + list(range(49, 55)) + list(range(62, 68))


def pushing(strict):
    from d3il_amd.envs.pushing import BlockPushVecEnv
    iq = G["avoiding__traj_last"].copy()
    ctx60 = np.load(os.path.join(ROOT, "d3il_amd", "data", "pushing_test_contexts.npy"))
    n = 120
    env = BlockPushVecEnv(n, device=0)
    env.set_option("solver_strict", strict)
    env.set_init_qpos(iq)
    env.reset(context=ctx60[np.arange(n) % 60])
    blob = blob_mod.load("pushing")
    o = Oracle(blob); o.env_start(iq)
    follow = {}
    for e in (0, 7, 33, 61, 90, 119):
        f = Oracle(blob); f.env_start(iq); f.push_reset(ctx60[e % 60]); follow[e] = f
    horizon = {e: {} for e in follow}
    des = env.robot_state()[:, :2].clone(); z = env.robot_state()[:, 2:3].clone()
    rng = np.random.default_rng(0)
    wp = wv = 0.0
    for t in range(90):
        o64 = env.obs.to(torch.float64)
        d = (o64[:, 2:4] if t < 45 else o64[:, 5:7]) - des
        nn = d.norm(dim=1, keepdim=True).clamp_min(1e-9)
        des = des + d / nn * torch.minimum(nn, torch.full_like(nn, 0.006))
        act = action(des, z)
        torch.cuda.synchronize()
        st0, fl0, sc0 = env.get_state()
        env.step(act)
        torch.cuda.synchronize()
        st1, fl1, sc1 = env.get_state()
        a = act.cpu().numpy()
        for e, f in follow.items():
            f.push_step(a[e])
            so, _ = f.push_state()
            err = max(np.abs(st1[P_POS, e] - so[P_POS]).max(), np.abs(st1[P_VEL, e] - so[P_VEL]).max())
            for thr in (1e-8, 1e-6, 1e-4):
                if err > thr and thr not in horizon[e]:
                    horizon[e][thr] = t
        if t < 20 or t % 2:
            continue
        for e in rng.choice(n, 6, replace=False):
            o.push_set_state(st0[:68, e], step=sc0[e], terminated=bool(fl0[e] & (1 << 12)), first_visit=int(fl0[e] & 7) - 1, ik_valid=bool(fl0[e] & (1 << 15)))
            o.push_step(a[e])
            so, _ = o.push_state()
            wp = max(wp, np.abs(st1[P_POS, e] - so[P_POS]).max()); wv = max(wv, np.abs(st1[P_VEL, e] - so[P_VEL]).max())
    bad = int((fl1 & ((1 << 16) | (1 << 18) | (1 << 19)) != 0).sum())
    print("pushing strict=%d: one-step max |dpos| %.3e  max |dvel| %.3e ; flagged envs %d" % (strict, wp, wv, bad))
    print("   horizon (first env step beyond threshold; 90 = never): " + "  ".join(
        "env %d: %s" % (e, "/".join(str(horizon[e].get(thr, 90)) for thr in (1e-8, 1e-6, 1e-4))) for e in follow))
    env.close()


def sorting(strict):
    from d3il_amd.envs.sorting import SortingVecEnv, sample_contexts
    NB = 4
    iq = G["sorting__traj_last"].copy()
    blob = blob_mod.load("sorting")
    n = 96
    ctx = sample_contexts(n, NB, seed=5)
    env = SortingVecEnv(n, device=0)
    env.set_option("solver_strict", strict)
    env.set_init_qpos(iq)
    env.reset(context=ctx)
    o = Oracle(blob); o.env_start(iq); o.sort_reset(ctx[0].reshape(NB, 7))

    def split(stc, e, orc):
        qp, qv = orc.state()
        cubes = np.concatenate([np.concatenate([qp[7 * b:7 * b + 7], qv[6 * b:6 * b + 6]]) for b in range(NB)])
        arm_q, arm_v = qp[7 * NB:7 * NB + 9], qv[6 * NB:6 * NB + 9]
        d = stc[42:42 + 13 * NB, e] - cubes
        vel = np.zeros(13 * NB, bool)
        for b in range(NB):
            vel[13 * b + 7:13 * b + 13] = True
        return max(np.abs(d[~vel]).max(), np.abs(stc[:9, e] - arm_q).max()), max(np.abs(d[vel]).max(), np.abs(stc[9:18, e] - arm_v).max())

    follow = {}
    for e in (0, 13, 31, 50, 77, 95):
        f = Oracle(blob); f.env_start(iq); f.sort_reset(ctx[e].reshape(NB, 7)); follow[e] = f
    horizon = {e: {} for e in follow}
    z = env.robot_state()[:, 2:3].clone()
    des = env.obs[:, :2].to(torch.float64).clone()
    g = torch.Generator(device="cpu").manual_seed(0)
    vel = torch.zeros(n, 2, dtype=torch.float64, device=des.device)
    wp = wv = 0.0
    for t in range(120):
        box = env.obs[:, 2:14].to(torch.float64).reshape(n, NB, 3)[:, :, :2]
        tgt = box[torch.arange(n), (torch.arange(n) + t // 30) % NB]
        d = tgt - des
        nn = d.norm(dim=1, keepdim=True)
        vel = 0.7 * vel + 0.3 * (d / nn.clamp_min(1e-9) * 0.006 + 0.002 * torch.randn(n, 2, generator=g, dtype=torch.float64).to(des.device))
        des = des + vel
        a = action(des, z)
        torch.cuda.synchronize()
        st, fl, sc = env.get_state()
        env.step(a)
        torch.cuda.synchronize()
        st2, fl2, sc2 = env.get_state()
        an = a.cpu().numpy()
        for e, f in follow.items():
            f.sort_step(an[e])
            ep, ev = split(st2, e, f)
            for thr in (1e-8, 1e-6, 1e-4):
                if max(ep, ev) > thr and thr not in horizon[e]:
                    horizon[e][thr] = t
        if t % 5 == 4:
            for e in [(t * 7 + k * 29) % n for k in range(4)]:
                o.sort_set_state(st[:, e], int(fl[e]), int(sc[e]))
                o.sort_step(an[e])
                ep, ev = split(st2, e, o)
                wp, wv = max(wp, ep), max(wv, ev)
    bad = int((fl2 & ((1 << 16) | (1 << 18) | (1 << 19)) != 0).sum())
    print("sorting strict=%d: one-step max |dpos| %.3e  max |dvel| %.3e ; flagged envs %d" % (strict, wp, wv, bad))
    print("   horizon (first env step beyond threshold; 120 = never): " + "  ".join(
        "env %d: %s" % (e, "/".join(str(horizon[e].get(thr, 120)) for thr in (1e-8, 1e-6, 1e-4))) for e in follow))
    env.close()


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "both"
    for strict in (0, 1):
        if which in ("pushing", "both"):
            pushing(strict)
        if which in ("sorting", "both"):
            sorting(strict)
