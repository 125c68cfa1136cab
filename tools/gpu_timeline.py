import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from d3il_amd.envs.avoiding import ObstacleAvoidanceVecEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
auto_reset = int(sys.argv[2]) if len(sys.argv) > 2 else 1
env = ObstacleAvoidanceVecEnv(n, device=0)
env.start(); env.reset(); env.policy_begin()
a = torch.zeros(n, 7, dtype=torch.float64, device=env.device)
env.set_timing(True)
for t in range(300):
    env.policy_action(42, 0, t, a)
    _, _, done, _ = env.step(a)
    ms = env.last_step_ms()
    if auto_reset:
        m = done.clone(); env.reset(m); env.policy_begin(m)
    if t % 10 == 0 or t in (248, 249, 250, 251, 252):
        st, fl, sc = env.get_state()
        print(t, "ms %.3f" % ms, "done", int(done.sum()), "tcp x [%.2f %.2f] y [%.2f %.2f]" % (st[25].min(), st[25].max(), st[26].min(), st[26].max()),
              "q range", np.abs(st[:7]).max(), "fing", st[7:9].min(), st[7:9].max(), "flags", [hex(x) for x in np.unique(fl.astype(np.int64) & 0xFFFFFE00)][:6])
