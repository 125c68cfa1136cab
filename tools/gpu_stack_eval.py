"""Stacking_Sim with the scripted pick-and-place policy on ALL 100 test contexts of the reference (one rollout each, 1000-step cap):
success rates for 1 / 2 / 3 boxes, order strings, engine flags.  A plausibility check of the grasp physics across contexts (the
reference's own learned policies reach 0.6 - 0.9 one-box success on this task) and the artifact profiles/r02/stacking_scripted_eval.json.
usage (GPU box): python tools/gpu_stack_eval.py [out.json]"""
import json
import os
import sys
import time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from d3il_amd.agents import ScriptedStackPolicy  # noqa: E402
from d3il_amd.controllers.scripted_stacking import build_trajectory  # noqa: E402
from d3il_amd.envs.stacking import CubeStackingVecEnv, mode_string  # noqa: E402
from d3il_amd.model import blob as blob_mod  # noqa: E402
from d3il_amd.simulation.stacking_sim import Stacking_Sim  # noqa: E402

n_ctx = 100
js = blob_mod.load_json("stacking")
sim = Stacking_Sim(seed=0, device="cuda:0", render=False, n_contexts=n_ctx, n_trajectories_per_context=1, max_steps_per_episode=1000)
probe = CubeStackingVecEnv(1, device=0)
q0, _, _ = probe.start(); probe.close()
t0 = time.time()
tables = [build_trajectory(js, q0, sim.test_contexts[c], speed=0.5) for c in range(n_ctx)]
t_ik = time.time() - t0
pol = ScriptedStackPolicy(tables, np.arange(n_ctx), device="cuda:0")
t0 = time.time()
succ, modes = sim.test_agent(pol)
torch.cuda.synchronize()
t_run = time.time() - t0
r = sim.last_rollout
fl = r["flags"].cpu().numpy()
mode = r["mode"].cpu().numpy()
orders = {}
for m in mode:
    s = mode_string(int(m)); orders[s] = orders.get(s, 0) + 1
out = {"contexts": n_ctx, "policy": "scripted pick-and-place (host IK once per context, speed 0.5: 888 steps)", "host_ik_s": round(t_ik, 1), "rollout_s": round(t_run, 2),
       "success_3_boxes": float(r["metrics"]["successes"]), "success_1_box": float(r["metrics"]["successes_1_box"]), "success_2_boxes": float(r["metrics"]["successes_2_boxes"]),
       "order_strings": orders,
       "flags": {"solver_fail": int(((fl >> 16) & 1).sum()), "contact_overflow": int(((fl >> 18) & 1).sum()), "off_table": int(((fl >> 19) & 1).sum()), "hand_near": int(((fl >> 20) & 1).sum())},
       "failed_contexts": [int(i) for i in np.nonzero(~r["success"].cpu().numpy())[0]]}
print(json.dumps(out, indent=1))
if len(sys.argv) > 1:
    with open(sys.argv[1], "w") as f:
        json.dump(out, f, indent=1)
