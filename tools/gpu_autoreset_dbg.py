import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from d3il_amd.envs.avoiding import ObstacleAvoidanceVecEnv
n = 512
iq = np.load("tests/golden/ref_offline_ik.npz")["avoiding__traj_last"]
A, B = ObstacleAvoidanceVecEnv(n, device=0), ObstacleAvoidanceVecEnv(n, device=0)
for env in (A, B):
    env.set_init_qpos(iq); env.reset(); env.policy_begin()
aa = torch.zeros(n, 7, dtype=torch.float64, device=A.device); ab = torch.zeros_like(aa)
counts = torch.zeros(2, dtype=torch.int64, device=A.device)
for t in range(270):
    A.policy_action(7, 100, t, aa); B.policy_action(7, 100, t, ab)
    _, _, da, _ = A.step(aa); _, _, db, _ = B.step(ab)
    torch.cuda.synchronize()
    sa0, fa0, ca0 = A.get_state(); sb0, fb0, cb0 = B.get_state()
    if not np.array_equal(sa0, sb0):
        print("t", t, "states differ BEFORE reset; envs", np.where((sa0 != sb0).any(0))[0][:10]); break
    nd = int(db.sum())
    A.auto_reset(counts); B.reset(db); B.policy_begin(db)
    torch.cuda.synchronize()
    sa, fa, ca = A.get_state(); sb, fb, cb = B.get_state()
    if nd or not np.array_equal(sa, sb):
        bad = np.where((sa != sb).any(0))[0]
        print("t", t, "done", nd, "state mismatch envs", bad[:10], "des equal", torch.equal(A.policy_des, B.policy_des), "flags eq", np.array_equal(fa, fb), "steps eq", np.array_equal(ca, cb), "done bufs", int(A.done.sum()), int(B.done.sum()), counts.tolist())
        if len(bad):
            e = bad[0]; print("  env", e, "fields", np.where(sa[:, e] != sb[:, e])[0], sa[:, e][:9], sb[:, e][:9]); break
