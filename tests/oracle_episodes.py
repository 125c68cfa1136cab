"""Whole evaluation episodes on the CPU oracle, one context per call (test infrastructure for tests/test_gpu_count_parity.py).

Every function mirrors the rollout loop of the corresponding batched Sim class (d3il_amd/simulation/*_sim.py, i.e. the reference's
simulation/pushing_sim.py:61-83, sorting_sim.py:118-133, stacking_sim.py:88-136) with the SAME closed-loop policy object, on a
batch of one, so that the integer outcome of an episode - success flag and behaviour-mode code of the step that returned `done` -
can be compared context by context with the device's.  `run_many` spreads the contexts over the host cores (spawned workers: the
parent process may hold an initialised HIP runtime)."""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _torch():
    import torch
    torch.set_num_threads(1)
    return torch


def pushing_episode(job):
    """job = (index, ctx14, init_qpos, max_steps, plan 0..3) -> (index, success, mode, steps, mean_distance)."""
    i, ctx, q0, max_steps, plan = job
    torch = _torch()
    from d3il_amd.agents import ScriptedGoalPushPolicy
    from d3il_amd.model import blob
    from oracle.oracle import Oracle
    o = Oracle(blob.load("pushing"))
    o.env_start(q0)
    obs = o.push_reset(ctx)
    s, _ = o.push_state()
    des, z = s[25:27].copy(), float(s[27])
    pol = ScriptedGoalPushPolicy("pushing", plan=[plan], device="cpu")
    info, t = dict(mode=-1, success=False, mean_distance=0.0), 0
    for t in range(max_steps):
        x = torch.as_tensor(np.concatenate([des, obs.astype(np.float64)])[None], dtype=torch.float64)
        des = des + pol.predict_batch(x)[0].numpy()
        obs, _, done, info = o.push_step(np.array([des[0], des[1], z, 0, 1, 0, 0]))
        if done:
            break
    return i, bool(info["success"]), int(info["mode"]), t + 1, float(info["mean_distance"])


def sorting_episode(job):
    """job = (index, ctx [4 x 7], init_qpos, max_steps) -> (index, success, mode code, steps)."""
    i, ctx, q0, max_steps = job
    torch = _torch()
    from d3il_amd.agents import ScriptedGoalPushPolicy
    from d3il_amd.model import blob
    from oracle.oracle import Oracle
    b = blob.load("sorting")
    o = Oracle(b)
    o.env_start(q0)
    obs = o.sort_reset(np.asarray(ctx).reshape(-1, 7))
    tcp = o.body(b.tcp_body)[0]
    z = float(tcp[2])
    # the harness latches the MEASURED tcp (f64), not its f32 observation (sorting_sim.py:120-121 -> env.robot_state())
    des = np.array([float(tcp[0]), float(tcp[1])])
    pol = ScriptedGoalPushPolicy("sorting", device="cpu")
    info, t = dict(mode=0, success=False), 0
    for t in range(max_steps):
        x = torch.as_tensor(np.concatenate([des, obs.astype(np.float64)])[None], dtype=torch.float64)
        des = des + pol.predict_batch(x)[0].numpy()
        obs, done, info = o.sort_step(np.array([des[0], des[1], z, 0, 1, 0, 0]))
        if done:
            break
    return i, bool(info["success"]), int(info["mode"]), t + 1


def stacking_episode(job):
    """job = (index, ctx21, init_qpos, max_steps, table [T, 8]) -> (index, success, mode string, steps).  The policy is the joint-space
    action table of the scripted pick-and-place; the loop keeps the f32 last-command arithmetic of stacking_sim.py:90-106."""
    i, ctx, q0, max_steps, table = job
    torch = _torch()
    from d3il_amd.agents import ScriptedStackPolicy
    from d3il_amd.model import blob
    from oracle.oracle import Oracle
    o = Oracle(blob.load("stacking"))
    o.env_start(q0)
    obs = o.stack_reset(ctx)
    pol = ScriptedStackPolicy([table], torch.zeros(1, dtype=torch.int64), device="cpu")
    pred = torch.as_tensor(o.stack_robot_state()[None], dtype=torch.float64).to(torch.float32)
    info, t = dict(mode="", success=False), 0
    for t in range(max_steps):
        obs20 = torch.cat((pred, torch.as_tensor(obs[None])), dim=1)
        out = pol.predict_batch(obs20).to(torch.float32)
        pred = torch.cat((out[:, :7] + obs20[:, :7], out[:, 7:8]), dim=1)
        obs, done, info = o.stack_step(pred[0].to(torch.float64).numpy())
        if done:
            break
    return i, bool(info["success"]), info["mode"], t + 1


def aligning_episode(job):
    """job = (index, ctx14, init_qpos, max_steps, inside) -> (index, success, mode, steps): the rollout loop of Aligning_Sim (aligning_sim.py:96-108)."""
    i, ctx, q0, max_steps, inside = job
    torch = _torch()
    from d3il_amd.agents import ScriptedAlignPolicy
    from d3il_amd.model import blob
    from oracle.oracle import Oracle
    b = blob.load("aligning")
    o = Oracle(b)
    o.env_start(q0)
    obs = o.align_reset(ctx)
    des = np.array(o.body(b.tcp_body)[0], dtype=np.float64)          # env.robot_state(): the measured TCP (f64), not its f32 observation
    pol = ScriptedAlignPolicy(inside=[bool(inside)], device="cpu")
    info, t = dict(mode=-1, success=False), 0
    for t in range(max_steps):
        x = torch.as_tensor(np.concatenate([des, obs.astype(np.float64)])[None], dtype=torch.float64)
        des = des + pol.predict_batch(x)[0].numpy()
        obs, _, done, info = o.align_step(np.concatenate([des, [0, 1, 0, 0]]))
        if done:
            break
    return i, bool(info["success"]), int(info["mode"]), t + 1


def n_workers() -> int:
    sys.path.insert(0, ROOT)
    from bench import _available_cores
    return max(1, _available_cores()[0])


def inserting_episode(job):
    """job = (index, ctx [3 x 7], init_qpos, waypoints [3 x 2], steps) -> (index, letters, code, success, first step with a letter or -1).  The action
    sequence is the open-loop waypoint walk of tests/test_gpu_count_parity.py (6 mm per step towards the current waypoint): a function of the job alone."""
    i, ctx, q0, way, steps = job
    from d3il_amd.model import blob
    from oracle.oracle import Oracle
    b = blob.load("inserting")
    o = Oracle(b)
    o.env_start(q0)
    obs = o.ins_reset(np.asarray(ctx, dtype=np.float64).reshape(21))
    tcp = o.body(b.tcp_body)[0]
    z = float(tcp[2])
    des = obs[:2].astype(np.float64)              # the harness latches the f32 observation of the TCP (like the device loop of the test)
    way = np.asarray(way, dtype=np.float64)
    wi, first, info = 0, -1, dict(n_mode=0, mode=0, success=False)
    for t in range(steps):
        d = way[wi] - des
        nn = float(np.sqrt((d * d).sum()))
        if nn < 1e-9 and wi < 2:
            wi += 1
            d = way[wi] - des
            nn = float(np.sqrt((d * d).sum()))
        des = des + d / max(nn, 1e-12) * min(nn, 0.006)
        obs, done, info = o.ins_step(np.array([des[0], des[1], z, 0, 1, 0, 0]))
        if info["n_mode"] and first < 0:
            first = t
    return i, int(info["n_mode"]), int(info["mode"]), bool(info["success"]), first


def run_many(fn, jobs):
    """Results of fn over the jobs, in job order, on all host cores this container may use."""
    import concurrent.futures as cf
    import multiprocessing as mp
    env_keep = {k: os.environ.get(k) for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "OMP_NUM_THREADS")}
    os.environ.update(HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="", OMP_NUM_THREADS="1")     # the workers never touch the GPU
    try:
        with cf.ProcessPoolExecutor(max_workers=min(n_workers(), len(jobs)), mp_context=mp.get_context("spawn")) as ex:
            out = list(ex.map(fn, jobs))
    finally:
        for k, v in env_keep.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return out
