"""solver_strict A/B of the whole-episode count tables: the device side of tests/test_gpu_count_parity.py's Pushing
tables under the production stopping rule and under the oracle's (solver_strict = 1), against the oracle outcomes stored by
tools/oracle_sensitivity.py (profiles/r04/oracle_sensitivity_*.json: run k = 0 is the unperturbed oracle episode, the other runs give the
set of outcomes the oracle itself reaches under 1e-12 m perturbations).

    python tools/gpu_count_strict.py [pushing] [pushing_sampled] [sorting]
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def device_rows(task, strict):
    from d3il_amd.agents import ScriptedGoalPushPolicy
    if task == "sorting":
        from d3il_amd.envs import sorting as envmod
        from d3il_amd.simulation.sorting_sim import Sorting_Sim
        cls = envmod.SortingVecEnv
    else:
        from d3il_amd.envs import pushing as envmod
        from d3il_amd.simulation.pushing_sim import Pushing_Sim
        cls = envmod.BlockPushVecEnv
    orig = cls.start

    def start(self):
        r = orig(self)
        self.set_option("solver_strict", strict)
        return r
    cls.start = start
    try:
        if task == "sorting":
            ctx = envmod.sample_contexts(60, 4, seed=0)
            sim = Sorting_Sim(seed=0, device="cuda:0", render=False, n_cores=1, n_contexts=60, n_trajectories_per_context=1, max_steps_per_episode=700, contexts=ctx)
            sim.test_agent(ScriptedGoalPushPolicy("sorting", device="cuda:0"))
        elif task == "pushing":
            sim = Pushing_Sim(seed=0, device="cuda:0", render=False, n_cores=1, n_contexts=60, n_trajectories_per_context=1, max_steps_per_episode=400)
            sim.test_agent(ScriptedGoalPushPolicy("pushing", plan=np.arange(60) % 4, device="cuda:0"))
        else:
            ctx = envmod.sample_contexts(120, seed=3)
            sim = Pushing_Sim(seed=0, device="cuda:0", render=False, n_cores=1, n_contexts=120, n_trajectories_per_context=1, max_steps_per_episode=400, contexts=ctx)
            sim.test_agent(ScriptedGoalPushPolicy("pushing", plan=np.arange(120) % 4, device="cuda:0"))
    finally:
        cls.start = orig
    r = sim.last_rollout
    return list(zip(r["success"].cpu().numpy().astype(bool).tolist(), r["mode"].cpu().numpy().tolist()))


def main():
    tasks = sys.argv[1:] or ["pushing", "pushing_sampled"]
    res = {}
    for task in tasks:
        p = os.path.join(ROOT, "profiles", "r04", "oracle_sensitivity_%s.json" % task)
        sens = json.load(open(p))["tasks"][task]
        outs = {int(k): v for k, v in sens["outcomes"].items()}
        for strict in (0, 1):
            rows = device_rows(task, strict)
            differ = [i for i, (s, m) in enumerate(rows) if [bool(s), int(m)] != outs[i][0][:2]]
            outside = [i for i, (s, m) in enumerate(rows) if [bool(s), int(m)] not in [o[:2] for o in outs[i]]]
            print("%s strict %d: device differs from the unperturbed oracle episode on %s; of these outside the oracle's own outcome set: %s (oracle-sensitive contexts: %s)" % (
                task, strict, differ, outside, sens["sensitive"]))
            res["%s_strict%d" % (task, strict)] = dict(differ=differ, outside_oracle_set=outside, device=[[bool(s), int(m)] for s, m in rows])
    out = os.path.join(ROOT, "gpurun_out", "count_strict_ab.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
