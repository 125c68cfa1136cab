"""Diagnostics: per-wave duration vs rare-path counts for late-episode steps (stats build)."""
import ctypes as C, os, sys, numpy as np, torch
os.environ.setdefault("D3IL_STATS_LIB", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from d3il_amd import capi
from d3il_amd.envs.avoiding import ObstacleAvoidanceVecEnv
n = 4096
env = ObstacleAvoidanceVecEnv(n, device=0)
L = capi.load()
env.start(); env.reset(); env.policy_begin()
a = torch.zeros(n, 7, dtype=torch.float64, device=env.device)
W = np.zeros((64, 10), dtype=np.uint64)
G = (C.c_uint64 * 32)()
L.d3il_debug_stats(G, 1)
names = ["jacobi", "deflate", "general", "newton_it", "finger", "contact", "ls_it", "serve_busy", "ik_busy", "ticks"]   # slot 7: busy ticks of the serving wave (three-wave kernel)
for t in range(260):
    env.policy_action(42, 0, t, a)
    L.d3il_debug_wave_stats(W.ctypes.data_as(C.c_void_p), 64, 1)
    _, _, done, _ = env.step(a)
    L.d3il_debug_wave_stats(W.ctypes.data_as(C.c_void_p), 64, 1)
    env.reset(done); env.policy_begin(done)
    if t in (30, 150, 200, 250):
        tot = np.maximum(W[:, 8], W[:, 9])
        order = np.argsort(tot)
        print("t", t, "busy ticks (100 MHz) per workgroup: ik min/med/max", np.min(W[:, 8]), int(np.median(W[:, 8])), np.max(W[:, 8]),
              "| physics min/med/max", np.min(W[:, 9]), int(np.median(W[:, 9])), np.max(W[:, 9]))
        for w in list(order[:2]) + list(order[30:32]) + list(order[-4:]):
            print("   wg %2d ik %7d phys %7d " % (w, W[w, 8], W[w, 9]) + " ".join("%s %d" % (names[i], W[w, i]) for i in (0, 1, 2, 3, 5, 6, 7)))
L.d3il_debug_stats(G, 0)
print("whole run (lane events): deflate solves %d, of which from a warm vector %d, Rayleigh-quotient steps %d (%.2f per deflate solve)" % (G[2], G[25], G[24], G[24] / max(1, G[2])))
