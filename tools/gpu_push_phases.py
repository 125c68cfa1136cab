"""Stats build: where the Pushing step spends its time (ticks of the 100 MHz wall clock, per workgroup)."""
import ctypes as C, os, sys, numpy as np, torch
os.environ["D3IL_STATS_LIB"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from d3il_amd import capi
from d3il_amd.envs.pushing import BlockPushVecEnv
n = 4096
ctx60 = np.load(os.path.join(ROOT, "d3il_amd", "data", "pushing_test_contexts.npy"))
ctx = ctx60[np.arange(n) % 60]
iq = np.load(os.path.join(ROOT, "tests", "golden", "ref_offline_ik.npz"))["avoiding__traj_last"]
env = BlockPushVecEnv(n, device=0)
L = capi.load()
env.set_init_qpos(iq); env.reset(context=ctx)
des = env.robot_state()[:, :2].clone(); z = env.robot_state()[:, 2:3].clone()
quat = torch.tensor([0.0, 1, 0, 0], dtype=torch.float64, device=env.device).expand(n, 4)
NW = 171
W = np.zeros((NW, 10), dtype=np.uint64)
names = ["c.setup", "c.A", "c.B", "c.C", "c.D", "c.E(ls)", "arm+prox", "general", "arm-dec", "cubes+int"]
for t in range(60):
    if t >= 12:
        o64 = env.obs.to(torch.float64); d = o64[:, 2:4] - des
        nn = d.norm(dim=1, keepdim=True).clamp_min(1e-9); des = des + d / nn * torch.minimum(nn, torch.full_like(nn, 0.006))
    torch.cuda.synchronize()
    L.d3il_debug_wave_stats(W.ctypes.data_as(C.c_void_p), NW, 1)
    env.step(torch.cat([des, z, quat], dim=1).contiguous())
    torch.cuda.synchronize()
    L.d3il_debug_wave_stats(W.ctypes.data_as(C.c_void_p), NW, 1)
    if t in (0, 8, 20, 45, 55):
        med = np.median(W.astype(np.float64), axis=0) / 100.0     # microseconds per env step
        print("t %2d  median us per step per workgroup: " % t + "  ".join("%s %.0f" % (names[i], med[i]) for i in range(10)))
