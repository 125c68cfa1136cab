"""Divergence-onset analysis of whole Sorting-4 episodes, device against oracle (the Sorting twin of tools/gpu_count_onset.py).

Contexts are drawn like tests/test_gpu_count_parity.py's (sample_contexts(60, 4, seed = 0)); the scripted push-over-the-edge policy runs
(a) on the device, state read back every env step; (b) free-running on the oracle; (c) as a one-step cross-check of the oracle along the
DEVICE trajectory (oracle loaded with the device state of step t, same action, compared with the device state of step t + 1).  Reported per
context: outcomes, device flag bits at the end (bit 19 = OFF_TABLE: a cube came near a pair the kernel does not evaluate), onset steps of the
free-running difference, and the largest one-step deviation with its step and the oracle's contact count / geom pairs there.

    python tools/gpu_count_onset_sorting.py --ctx 52,51,53,0 --out gpurun_out/onset_sorting.json
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NB = 4


def dev_layout(o):
    """Oracle state in the device layout rows 0..17 (arm q[9] v[9]) and 42..42 + 13 NB (cubes: pos3 quat4 vel6)."""
    qp, qv = o.state()
    cubes = np.concatenate([np.concatenate([qp[7 * b:7 * b + 7], qv[6 * b:6 * b + 6]]) for b in range(NB)])
    return np.concatenate([qp[7 * NB:7 * NB + 9], qv[6 * NB:6 * NB + 9]]), cubes


def split_err(col, arm, cubes):
    d = col[42:42 + 13 * NB] - cubes
    vel = np.zeros(13 * NB, bool)
    for b in range(NB):
        vel[13 * b + 7:13 * b + 13] = True
    dp = max(float(np.abs(d[~vel]).max()), float(np.abs(col[:9] - arm[:9]).max()))
    dv = max(float(np.abs(d[vel]).max()), float(np.abs(col[9:18] - arm[9:]).max()))
    return dp, dv


def oracle_free_run(job):
    i, ctx, q0, max_steps = job
    torch.set_num_threads(1)
    from d3il_amd.agents import ScriptedGoalPushPolicy
    from d3il_amd.model import blob
    from oracle.oracle import Oracle
    b = blob.load("sorting")
    o = Oracle(b)
    o.env_start(q0)
    obs = o.sort_reset(np.asarray(ctx).reshape(-1, 7))
    tcp = o.body(b.tcp_body)[0]
    z = float(tcp[2])
    des = np.array([float(tcp[0]), float(tcp[1])])
    pol = ScriptedGoalPushPolicy("sorting", device="cpu")
    traj, info, t = [dev_layout(o)], dict(mode=0, success=False), 0
    for t in range(max_steps):
        x = torch.as_tensor(np.concatenate([des, obs.astype(np.float64)])[None], dtype=torch.float64)
        des = des + pol.predict_batch(x)[0].numpy()
        obs, done, info = o.sort_step(np.array([des[0], des[1], z, 0, 1, 0, 0]))
        traj.append(dev_layout(o))
        if done:
            break
    return i, traj, bool(info["success"]), int(info["mode"]), t + 1


def oracle_one_step(job):
    i, q0, st, fl, sc, act = job
    from d3il_amd.model import blob
    from oracle.oracle import Oracle
    o = Oracle(blob.load("sorting"))
    o.env_start(q0)
    rows = []
    for t in range(len(act)):
        o.sort_set_state(st[t], int(fl[t]), int(sc[t]))
        _, _, info = o.sort_step(act[t])
        arm, cubes = dev_layout(o)
        dp, dv = split_err(st[t + 1], arm, cubes)
        con = o.contacts()
        pairs = sorted({(int(a), int(b)) for a, b in con[:, 8:10]}) if len(con) else []
        rows.append((dp, dv, int(len(con)), pairs, int(info["mode"]), bool(info["success"])))
    return i, rows


def main():
    args = sys.argv[1:]
    ctx_ids, out, max_steps, strict = [52, 51, 53, 0], None, 700, 0
    it = iter(args)
    for a in it:
        if a == "--ctx":
            ctx_ids = [int(x) for x in next(it).split(",")]
        elif a == "--out":
            out = next(it)
        elif a == "--strict":
            strict = int(next(it))
    from d3il_amd.agents import ScriptedGoalPushPolicy
    from d3il_amd.envs.sorting import SortingVecEnv, sample_contexts
    from tests import oracle_episodes as oe
    ctx = sample_contexts(60, 4, seed=0)[ctx_ids]
    n = len(ctx_ids)
    dev = torch.device("cuda:0")
    env = SortingVecEnv(n, device=dev, render=False, max_steps_per_episode=max_steps, num_boxes=4)
    q0 = env.start()[0]
    env.set_option("solver_strict", strict)
    obs = env.reset(random=False, context=ctx)
    pol = ScriptedGoalPushPolicy("sorting", device="cuda:0")
    rs = env.robot_state().clone()
    fixed_z, des_xy = rs[:, 2:3].clone(), rs[:, :2].clone()
    quat = torch.tensor([0.0, 1.0, 0.0, 0.0], dtype=torch.float64, device=dev).expand(n, 4)
    finished = torch.zeros(n, dtype=torch.bool, device=dev)
    torch.cuda.synchronize()
    st, fl, sc = env.get_state()
    S, F, C, A = [st.copy()], [fl.copy()], [sc.copy()], []
    outcome = [None] * n
    for t in range(max_steps):
        obs_in = torch.cat((des_xy, obs.to(torch.float64)), dim=1)
        delta = pol.predict_batch(obs_in).to(device=dev, dtype=torch.float64).reshape(n, 2)
        des_xy = torch.where(finished.unsqueeze(1), des_xy, delta + obs_in[:, :2])
        action = torch.cat((des_xy, fixed_z, quat), dim=1).contiguous()
        obs, _, done, info = env.step(action)
        torch.cuda.synchronize()
        st, fl, sc = env.get_state()
        S.append(st.copy()); F.append(fl.copy()); C.append(sc.copy()); A.append(action.cpu().numpy().copy())
        d = done.bool().cpu().numpy()
        for e in range(n):
            if d[e] and outcome[e] is None:
                outcome[e] = (bool(info["success"][e]), int(info["mode"][e]), t + 1, int(fl[e]))
        finished |= done.bool()
        if bool(finished.all()):
            break
    env.close()
    S, F, C, A = np.array(S), np.array(F), np.array(C), np.array(A)
    free = oe.run_many(oracle_free_run, [(e, ctx[e], q0, max_steps) for e in range(n)])
    T_dev = [outcome[e][2] for e in range(n)]
    one = oe.run_many(oracle_one_step, [(e, q0, S[:T_dev[e] + 1, :, e], F[:T_dev[e] + 1, e], C[:T_dev[e] + 1, e], A[:T_dev[e], e]) for e in range(n)])
    report = dict(task="sorting", strict=strict, contexts=[])
    print("ctx | device (succ, mode, steps, flags) | oracle (succ, mode, steps) | onset 1e-9 / 1e-6 / 1e-4 | one-step max |dpos| @t (ncon) | max |dvel| @t (ncon) | steps over 2e-8 / 2e-6 | one-step (mode, success) mismatches")
    for e in range(n):
        _, traj, succ_o, mode_o, T_o = free[e]
        T = min(T_dev[e], T_o)
        diff = np.array([split_err(S[t, :, e], *traj[t])[0] for t in range(1, T + 1)])
        onset = {}
        for thr in (1e-9, 1e-6, 1e-4):
            w = np.nonzero(diff > thr)[0]
            onset[thr] = int(w[0]) + 1 if len(w) else None
        rows = one[e][1]
        dp = np.array([r[0] for r in rows]); dv = np.array([r[1] for r in rows]); nc = [r[2] for r in rows]
        tp, tv = int(dp.argmax()), int(dv.argmax())
        over = (int((dp > 2e-8).sum()), int((dv > 2e-6).sum()))
        spikes = [(t, float(dp[t]), float(dv[t]), nc[t], rows[t][3]) for t in np.nonzero((dp > 2e-8) | (dv > 2e-6))[0][:12]]
        rec = dict(ctx=ctx_ids[e], device=[outcome[e][0], outcome[e][1], outcome[e][2], hex(outcome[e][3])], oracle=[succ_o, mode_o, T_o], onset={str(k): v for k, v in onset.items()},
                   one_step_max_dpos=float(dp.max()), at_pos=tp, ncon_pos=nc[tp], pairs_pos=rows[tp][3], one_step_max_dvel=float(dv.max()), at_vel=tv, ncon_vel=nc[tv],
                   pairs_vel=rows[tv][3], steps_over_bounds=list(over), spikes=spikes, one_step_dpos=dp.tolist(), one_step_dvel=dv.tolist(), ncon=nc, free_running_diff=diff.tolist())
        report["contexts"].append(rec)
        print("%3d | %s | %s | %s / %s / %s | %.2e @%d (%d) | %.2e @%d (%d) | %d / %d" % (
            ctx_ids[e], (outcome[e][0], outcome[e][1], outcome[e][2], hex(outcome[e][3])), (succ_o, mode_o, T_o), onset[1e-9], onset[1e-6], onset[1e-4], dp.max(), tp, nc[tp], dv.max(), tv, nc[tv], over[0], over[1]))
        for s in spikes:
            print("      spike at step %d: |dpos| %.2e |dvel| %.2e, oracle contacts %d pairs %s" % s)
    if out:
        os.makedirs(os.path.dirname(os.path.join(ROOT, out)) or ".", exist_ok=True)
        with open(os.path.join(ROOT, out), "w") as f:
            json.dump(report, f)


if __name__ == "__main__":
    main()
