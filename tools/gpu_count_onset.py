"""Divergence-onset analysis of whole Pushing episodes, device against oracle.

For the given contexts the closed-loop scripted policy of tests/test_gpu_count_parity.py (plan = context % 4) runs

  (a) on the device (BlockPushVecEnv = the C ABI), the full state of every environment read back after every env step;
  (b) free-running on the CPU oracle (tests/oracle_episodes.py's loop), state recorded after every env step;
  (c) as a ONE-STEP cross-check along the device trajectory: at every env step t the oracle is loaded with the DEVICE state of
      step t, takes the device's action of that step, and its result is compared with the device state of step t + 1.

Reported per context: the first env step at which |device - oracle| (free-running, position rows) exceeds 1e-9 / 1e-6 / 1e-4 (the
onset), the outcome (success, mode) of either side, and over the WHOLE episode the largest one-step deviation of (c) in positions and
velocities with the step at which it occurs, the oracle's contact count and pair set there, and the number of steps whose one-step
deviation exceeds the bounds asserted by test_one_step_parity_from_mid_episode_states (2e-8 / 2e-6).  A discrete disagreement between
the two collision / solver paths (a contact one side has and the other has not, a different reference face) is a one-step deviation
of contact-force size (>= 1e-5 in velocity); round-off growth of a chaotic contact system shows one-step deviations at the
conditioning level everywhere and an exponentially growing free-running difference.

    python tools/gpu_count_onset.py --ctx 10,26,34,54,6,22,2,14 [--strict 1] [--sampled] --out gpurun_out/onset_pushing.json
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
POS = list(range(0, 9)) + list(range(25, 28)) + list(range(42, 49)) + list(range(55, 62))
VEL = list(range(9, 18)) + list(range(49, 55)) + list(range(62, 68))


def oracle_free_run(job):
    i, ctx, q0, max_steps, plan = job
    torch.set_num_threads(1)
    from d3il_amd.agents import ScriptedGoalPushPolicy
    from d3il_amd.model import blob
    from oracle.oracle import Oracle
    o = Oracle(blob.load("pushing"))
    o.env_start(q0)
    obs = o.push_reset(ctx)
    s, _ = o.push_state()
    des, z = s[25:27].copy(), float(s[27])
    pol = ScriptedGoalPushPolicy("pushing", plan=[plan], device="cpu")
    states, info, t = [s.copy()], dict(mode=-1, success=False), 0
    for t in range(max_steps):
        x = torch.as_tensor(np.concatenate([des, obs.astype(np.float64)])[None], dtype=torch.float64)
        des = des + pol.predict_batch(x)[0].numpy()
        obs, _, done, info = o.push_step(np.array([des[0], des[1], z, 0, 1, 0, 0]))
        states.append(o.push_state()[0].copy())
        if done:
            break
    return i, np.array(states), bool(info["success"]), int(info["mode"]), t + 1


def oracle_one_step(job):
    """(c): job = (i, q0, device states [T + 1, 68], flags [T + 1], steps [T + 1], actions [T, 7]) -> per step |dpos|, |dvel|, ncon, pairs."""
    i, q0, st, fl, sc, act = job
    from d3il_amd.model import blob
    from oracle.oracle import Oracle
    o = Oracle(blob.load("pushing"))
    o.env_start(q0)
    rows = []
    for t in range(len(act)):
        o.push_set_state(st[t], step=int(sc[t]), terminated=bool(fl[t] & (1 << 12)), first_visit=int(fl[t] & 7) - 1, ik_valid=bool(fl[t] & (1 << 15)))
        o.push_step(act[t])
        so, _ = o.push_state()
        con = o.contacts()
        pairs = sorted({(int(a), int(b)) for a, b in con[:, 8:10]}) if len(con) else []
        rows.append((float(np.abs(st[t + 1][POS] - so[POS]).max()), float(np.abs(st[t + 1][VEL] - so[VEL]).max()), int(len(con)), pairs))
    return i, rows


def main():
    args = sys.argv[1:]
    ctx_ids, strict, out, sampled, max_steps = [10, 26, 34, 54, 6, 22, 2, 14], 0, None, False, 400
    it = iter(args)
    for a in it:
        if a == "--ctx":
            ctx_ids = [int(x) for x in next(it).split(",")]
        elif a == "--strict":
            strict = int(next(it))
        elif a == "--out":
            out = next(it)
        elif a == "--sampled":
            sampled = True
    from d3il_amd.agents import ScriptedGoalPushPolicy
    from d3il_amd.envs.pushing import BlockPushVecEnv, sample_contexts
    from d3il_amd.simulation.pushing_sim import load_test_contexts
    from tests import oracle_episodes as oe
    all_ctx = sample_contexts(120, seed=3) if sampled else load_test_contexts()
    ctx = all_ctx[ctx_ids]
    n = len(ctx_ids)
    dev = torch.device("cuda:0")
    env = BlockPushVecEnv(n, device=dev, render=False, max_steps_per_episode=max_steps)
    q0 = env.start()[0]
    env.set_option("solver_strict", strict)
    obs = env.reset(random=False, context=ctx)
    pol = ScriptedGoalPushPolicy("pushing", plan=np.array(ctx_ids) % 4, device="cuda:0")
    rs = env.robot_state().clone()
    fixed_z, des_xy = rs[:, 2:3].clone(), rs[:, :2].clone()
    quat = torch.tensor([0.0, 1.0, 0.0, 0.0], dtype=torch.float64, device=dev).expand(n, 4)
    finished = torch.zeros(n, dtype=torch.bool, device=dev)
    torch.cuda.synchronize()
    st, fl, sc = env.get_state()
    S, F, C, A = [st[:68].copy()], [fl.copy()], [sc.copy()], []
    outcome = [None] * n
    for t in range(max_steps):
        obs10 = torch.cat((des_xy, obs.to(torch.float64)), dim=1)
        delta = pol.predict_batch(obs10).to(device=dev, dtype=torch.float64).reshape(n, 2)
        des_xy = torch.where(finished.unsqueeze(1), des_xy, delta + obs10[:, :2])
        action = torch.cat((des_xy, fixed_z, quat), dim=1).contiguous()
        obs, _, done, info = env.step(action)
        torch.cuda.synchronize()
        st, fl, sc = env.get_state()
        S.append(st[:68].copy()); F.append(fl.copy()); C.append(sc.copy()); A.append(action.cpu().numpy().copy())
        d = done.bool().cpu().numpy()
        for e in range(n):
            if d[e] and outcome[e] is None:
                outcome[e] = (bool(info["success"][e]), int(info["mode"][e]), t + 1)
        finished |= done.bool()
        if bool(finished.all()):
            break
    env.close()
    S, F, C, A = np.array(S), np.array(F), np.array(C), np.array(A)
    free = oe.run_many(oracle_free_run, [(e, ctx[e], q0, max_steps, ctx_ids[e] % 4) for e in range(n)])
    T_dev = [outcome[e][2] for e in range(n)]
    one = oe.run_many(oracle_one_step, [(e, q0, S[:T_dev[e] + 1, :, e], F[:T_dev[e] + 1, e], C[:T_dev[e] + 1, e], A[:T_dev[e], e]) for e in range(n)])
    report = dict(task="pushing_sampled" if sampled else "pushing", strict=strict, contexts=[])
    print("ctx plan | device (succ, mode, steps) | oracle (succ, mode, steps) | onset 1e-9 / 1e-6 / 1e-4 | one-step max |dpos| @t (ncon) | max |dvel| @t (ncon) | steps over 2e-8 / 2e-6")
    for e in range(n):
        _, so, succ_o, mode_o, T_o = free[e]
        T = min(T_dev[e], T_o)
        diff = np.abs(S[1:T + 1, :, e][:, POS] - so[1:T + 1][:, POS]).max(axis=1)
        onset = {}
        for thr in (1e-9, 1e-6, 1e-4):
            w = np.nonzero(diff > thr)[0]
            onset[thr] = int(w[0]) + 1 if len(w) else None
        rows = one[e][1]
        dp = np.array([r[0] for r in rows]); dv = np.array([r[1] for r in rows]); nc = [r[2] for r in rows]
        tp, tv = int(dp.argmax()), int(dv.argmax())
        over = (int((dp > 2e-8).sum()), int((dv > 2e-6).sum()))
        # growth factor per env step of the free-running difference between the 1e-9 and 1e-4 onsets
        growth = None
        if onset[1e-9] and onset[1e-4] and onset[1e-4] > onset[1e-9]:
            growth = float(10 ** (5.0 / (onset[1e-4] - onset[1e-9])))
        rec = dict(ctx=ctx_ids[e], plan=ctx_ids[e] % 4, device=list(outcome[e]), oracle=[succ_o, mode_o, T_o], onset={str(k): v for k, v in onset.items()},
                   one_step_max_dpos=float(dp.max()), at_pos=tp, ncon_pos=nc[tp], pairs_pos=rows[tp][3], one_step_max_dvel=float(dv.max()), at_vel=tv, ncon_vel=nc[tv],
                   pairs_vel=rows[tv][3], steps_over_bounds=list(over), growth_per_step=growth,
                   one_step_dpos=dp.tolist(), one_step_dvel=dv.tolist(), ncon=nc, free_running_diff=diff.tolist())
        report["contexts"].append(rec)
        print("%3d  %d | %s | %s | %s / %s / %s | %.2e @%d (%d) | %.2e @%d (%d) | %d / %d%s" % (
            ctx_ids[e], ctx_ids[e] % 4, outcome[e], (succ_o, mode_o, T_o), onset[1e-9], onset[1e-6], onset[1e-4], dp.max(), tp, nc[tp], dv.max(), tv, nc[tv], over[0], over[1],
            "" if growth is None else " | x%.2f per step" % growth))
    if out:
        os.makedirs(os.path.dirname(os.path.join(ROOT, out)) or ".", exist_ok=True)
        with open(os.path.join(ROOT, out), "w") as f:
            json.dump(report, f)


if __name__ == "__main__":
    main()
