"""Outcome sensitivity of whole evaluation episodes ON THE CPU ORACLE ALONE (no GPU).

For every context of the count-parity tables (tests/test_gpu_count_parity.py) the oracle episode is run K + 1 times: once on
the context as given and K times with cube 0's x moved by k * 1e-12 m (k = 1 .. K) - a perturbation five orders of magnitude
below the f32 observations the policy sees and four below the one-step agreement of any two f64 implementations of the
soft-contact step.  A context whose (success, mode) outcome is not unanimous over these runs is one whose
integer outcome is not a function of the context at f64 resolution: there, "device == oracle" cannot be asserted of ANY second
implementation, the oracle re-run on a different machine's libm included.  The table lists those contexts next to the
contexts on which the device differed (profiles/r03/count_parity_*.json), so the reader sees whether the latter are a subset.

    python tools/oracle_sensitivity.py [pushing] [pushing_sampled] [sorting] [--k 6] [--eps 1e-12] [--out profiles/r04/oracle_sensitivity.json]
                                       [--fixture tests/golden/oracle_outcome_sets.json]   (the committed fixture: k 24, eps 1e-12; Sorting eps 1e-10)
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _q0(task):
    """init_qpos of the task's start pose: the offline IK of the reference's env.start() (host code, no GPU; what
    ObstacleAvoidanceVecEnv.start does, d3il_amd/envs/avoiding.py)."""
    from d3il_amd.controllers.offline_ik import offline_ik
    from d3il_amd.kinematics import UrdfChain
    from d3il_amd.model import blob
    js = blob.load_json(task)
    c, tc = js["controller"], js["task_const"]
    target = list(tc["init_end_eff_pos"]) + list(tc["init_end_eff_quat"])
    q, _, _ = offline_ik(UrdfChain(js["urdf_chain"]), c["default_qpos"], target, np.array(c["joint_pos_min"]), np.array(c["joint_pos_max"]))
    return q


RANDOM = False


def jobs_for(task, k_pert, eps):
    from tests import oracle_episodes as oe
    if task in ("pushing", "pushing_sampled"):
        from d3il_amd.simulation.pushing_sim import load_test_contexts
        from d3il_amd.envs.pushing import sample_contexts
        ctx = load_test_contexts() if task == "pushing" else sample_contexts(120, seed=3)
        q0 = _q0("pushing")
        jobs = []
        for i in range(len(ctx)):
            for k in range(k_pert + 1):
                c = np.array(ctx[i], dtype=np.float64).copy()
                c[0] += k * eps
                jobs.append((i * 100 + k, c, q0, 400, i % 4))
        return oe.pushing_episode, jobs, len(ctx)
    if task == "sorting":
        from d3il_amd.envs.sorting import sample_contexts
        ctx = sample_contexts(60, 4, seed=0)
        q0 = _q0("sorting")
        jobs = []
        for i in range(len(ctx)):
            for k in range(k_pert + 1):
                c = np.array(ctx[i], dtype=np.float64).reshape(-1, 7).copy()
                if RANDOM:      # every cube's x and y by an independent uniform draw in (-eps, eps): directions like an implementation's round-off, not one axis
                    c[:, :2] += (np.random.default_rng(1000 * i + k).uniform(-eps, eps, size=(c.shape[0], 2)) if k else 0.0)
                else:
                    c[:, 0] += k * eps * np.arange(1, c.shape[0] + 1)      # every cube moves, by a different multiple
                jobs.append((i * 100 + k, c, q0, 700))
        return oe.sorting_episode, jobs, len(ctx)
    raise SystemExit("unknown task " + task)


def main():
    args = sys.argv[1:]
    k_pert, eps, out, fixture = 6, 1e-12, None, None
    tasks = []
    it = iter(args)
    for a in it:
        if a == "--k":
            k_pert = int(next(it))
        elif a == "--eps":
            eps = float(next(it))
        elif a == "--out":
            out = next(it)
        elif a == "--fixture":
            fixture = next(it)
        elif a == "--random":
            global RANDOM
            RANDOM = True
        else:
            tasks.append(a)
    tasks = tasks or ["pushing", "pushing_sampled", "sorting"]
    from tests import oracle_episodes as oe
    result = dict(k=k_pert, eps=eps, tasks={})
    for task in tasks:
        fn, jobs, nctx = jobs_for(task, k_pert, eps)
        res = oe.run_many(fn, jobs)
        rows = {}
        for r in res:
            i, k = divmod(r[0], 100)
            rows.setdefault(i, []).append((k, bool(r[1]), int(r[2]), int(r[3])))
        sens = []
        table = {}
        for i in range(nctx):
            outs = sorted(rows[i])
            table[i] = [[s, m, t] for _, s, m, t in outs]
            if len({(s, m) for _, s, m, _ in outs}) > 1:
                sens.append(i)
        prev = None
        p = os.path.join(ROOT, "profiles", "r03", "count_parity_%s.json" % task)
        if os.path.exists(p):
            prev = json.load(open(p))["differing"]
        result["tasks"][task] = dict(contexts=nctx, sensitive=sens, device_differed_r03=prev,
                                     subset=(None if prev is None else sorted(set(prev) - set(sens)) == []), outcomes=table)
        print("%s: %d contexts, %d with a non-unanimous oracle outcome under %d perturbations of %.0e m: %s" % (task, nctx, len(sens), k_pert, eps, sens))
        print("   contexts on which the device differed in round 3: %s" % (prev,))
        for i in sens:
            print("   ctx %3d: %s" % (i, ["%s/%d@%d" % ("S" if s else "F", m, t) for s, m, t in table[i]]))
    if out:
        os.makedirs(os.path.dirname(os.path.join(ROOT, out)), exist_ok=True)
        with open(os.path.join(ROOT, out), "w") as f:
            json.dump(result, f, indent=1)
    if fixture:
        # the compact form tests/test_gpu_count_parity.py reads: per task and context the DISTINCT (success, mode) outcomes of the oracle
        # under the perturbations; one entry = the outcome is decided at f64 resolution, several = it is not
        path = os.path.join(ROOT, fixture)
        fx = json.load(open(path)) if os.path.exists(path) else {}
        for task, r in result["tasks"].items():
            fx[task] = dict(k=k_pert, eps=eps, perturbation=("x, y of every cube + independent uniform draws in (-eps, eps), k draws (rng seed 1000 ctx + j)" if RANDOM else
                                          "cube x positions of the context + j * eps (Sorting: cube b by (b + 1) j eps), j = 0 .. k"),
                            outcomes={str(i): sorted({(bool(s), int(m)) for s, m, _ in rows}) for i, rows in r["outcomes"].items()})
        with open(path, "w") as f:
            json.dump(fx, f, indent=0, sort_keys=True)


if __name__ == "__main__":
    main()
