"""Stacking: dump (state before, action, state after) triples of device env steps along scripted pick-and-place rollouts, for
replay against the host build / oracle on a machine without a GPU.  Run on the GPU box; writes gpurun_out/stack_dump.npz."""
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from d3il_amd.controllers.scripted_stacking import build_trajectory  # noqa: E402
from d3il_amd.envs.stacking import CubeStackingVecEnv  # noqa: E402
from d3il_amd.model import blob as blob_mod  # noqa: E402

js = blob_mod.load_json("stacking")
ctx100 = np.load(os.path.join(ROOT, "d3il_amd", "data", "stacking_test_contexts.npy"))
ids = [1, 5, 17, 60, 77, 93]
n = len(ids)
NSUB = int(sys.argv[1]) if len(sys.argv) > 1 else 30       # 1: one physics sub-step per device call (the action repeats 30 times)
env = CubeStackingVecEnv(n, device=0, n_substeps=NSUB, max_steps_per_episode=1000000)
q0, _, _ = env.start()
env.reset(context=ctx100[ids])
trajs = [build_trajectory(js, q0, ctx100[i], n_boxes=2, speed=1.0) for i in ids]
T = min(len(t) for t in trajs)
S0, F0, C0, A, S1, F1 = [], [], [], [], [], []
REP = 30 // NSUB
for tt in range(T * REP):
    t = tt // REP
    if NSUB == 1 and not (36 <= t < 60 or 110 <= t < 130):     # sub-step resolution around the first grasp and the first release only
        act = np.stack([trajs[k][t] for k in range(n)])
        env.step(torch.as_tensor(act, dtype=torch.float64, device=env.device).contiguous())
        continue
    act = np.stack([trajs[k][t] for k in range(n)])
    torch.cuda.synchronize()
    st0, fl0, sc0 = env.get_state()
    env.step(torch.as_tensor(act, dtype=torch.float64, device=env.device).contiguous())
    torch.cuda.synchronize()
    st1, fl1, sc1 = env.get_state()
    S0.append(st0.copy()); F0.append(fl0.copy()); C0.append(sc0.copy()); A.append(act); S1.append(st1.copy()); F1.append(fl1.copy())
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "stack_dump.npz" if NSUB == 30 else "stack_dump_sub%d.npz" % NSUB), s0=np.array(S0), f0=np.array(F0), c0=np.array(C0), a=np.array(A), s1=np.array(S1), f1=np.array(F1), q0=q0, ids=np.array(ids))
print("dumped", len(S0), "steps x", n, "envs")
