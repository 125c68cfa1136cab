"""Diagnostics: per-step lane/wave counts of the rare paths (needs libd3il_rollout_stats.so, D3IL_STATS_LIB=1)."""
import ctypes as C, os, sys, numpy as np, torch
os.environ["D3IL_STATS_LIB"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from d3il_amd import capi
from d3il_amd.envs.avoiding import ObstacleAvoidanceVecEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = ObstacleAvoidanceVecEnv(n, device=0)
L = capi.load()
env.start(); env.reset(); env.policy_begin()
a = torch.zeros(n, 7, dtype=torch.float64, device=env.device)
buf = (C.c_uint64 * 32)()
L.d3il_debug_stats(buf, 1)
names = ["jacobi", "deflate", "general", "newton_it", "finger", "contact", "ls_it", "substep"]
env.set_timing(True)
for t in range(300):
    env.policy_action(42, 0, t, a)
    _, _, done, _ = env.step(a)
    ms = env.last_step_ms()
    L.d3il_debug_stats(buf, 1)
    m = done.clone(); env.reset(m); env.policy_begin(m)
    L.d3il_debug_stats((C.c_uint64 * 32)(), 1)
    if t % 20 == 0 or t == 299 or t in (251, 255):
        v = list(buf)
        print(t, "ms %.2f" % ms, " ".join("%s %d/%d" % (names[i], v[2 * i], v[2 * i + 1]) for i in range(8)))
