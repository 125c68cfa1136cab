"""Stats build: where the Sorting step spends its time (ticks of the 100 MHz wall clock, per workgroup): the phases of
the physics wave (gen_kernels.h) and, inside the joint island solves, the solver passes (gen_step.h)."""
import ctypes as C, os, sys, numpy as np, torch
os.environ["D3IL_STATS_LIB"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from d3il_amd import capi
from d3il_amd.envs.sorting import SortingVecEnv, sample_contexts
PER_WAVE = "--per-wave" in sys.argv      # library built with python -m d3il_amd.build --per-wave: timer rows are waves, row 4 b + 3 = barrier waits of the waves of workgroup b
if PER_WAVE:
    sys.argv.remove("--per-wave")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
iq = np.load(os.path.join(ROOT, "tests", "golden", "ref_offline_ik.npz"))["sorting__traj_last"]
env = SortingVecEnv(n, device=0)
L = capi.load()
env.set_init_qpos(iq); env.reset(context=sample_contexts(n, 4, seed=1))
des = env.robot_state()[:, :2].clone(); z = env.robot_state()[:, 2:3].clone()
quat = torch.tensor([0.0, 1, 0, 0], dtype=torch.float64, device=env.device).expand(n, 4)
NW = (n + 15) // 16 * (4 if PER_WAVE else 1)
W = np.zeros((NW, 10), dtype=np.uint64)
CN = np.zeros((NW, 8), dtype=np.uint64)
cnames = ["newton its", "own-contact trips", "partner trips", "ls its", "ls contact trips", "jp trips", "sub-steps with a solve", "generic solves"]
names = ["p1.arm", "p2.statics", "p3.bb+rod+reduce", "t.setup", "t.grad+H", "t.elim+solve", "t.jp", "t.linesearch+step", "p4.tree(total)", "p4.generic"]      # slots 3..7: inside the tree solver (gen_tree.h; the rare generic solves add to them)
TS = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [20, 39, 55]
for t in range(max(TS) + 1):
    box = env.obs[:, 2:4].to(torch.float64)
    if t >= 12:
        aligned = ((des[:, 0] - box[:, 0]).abs() < 0.008) & (des[:, 1] < box[:, 1] - 0.02)
        target = torch.where(aligned[:, None], torch.stack([box[:, 0], torch.full_like(box[:, 0], 0.36)], 1), box + torch.tensor([0.0, -0.06], dtype=torch.float64, device=box.device))
        d = target - des; nn = d.norm(dim=1, keepdim=True)
        des = des + d / nn.clamp_min(1e-9) * torch.minimum(nn, torch.full_like(nn, 0.006))
    torch.cuda.synchronize()
    L.d3il_debug_wave_stats(W.ctypes.data_as(C.c_void_p), NW, 1)
    L.d3il_debug_wave_counts(CN.ctypes.data_as(C.c_void_p), NW, 1)
    env.step(torch.cat([des, z, quat], dim=1).contiguous())
    torch.cuda.synchronize()
    L.d3il_debug_wave_stats(W.ctypes.data_as(C.c_void_p), NW, 1)
    L.d3il_debug_wave_counts(CN.ctypes.data_as(C.c_void_p), NW, 1)
    if t in TS:
        Wf = W.astype(np.float64)
        Wf /= (100.0 if PER_WAVE else 200.0)      # per workgroup: the mean of its two physics waves
        if PER_WAVE:      # per PHYSICS wave: busy microseconds (timed phases), barrier wait; per workgroup: the slower wave's busy + wait ~ the workgroup's duration
            R = Wf.reshape(-1, 4, 10)
            busy = R[:, 1:3, [0, 1, 2, 8, 9]].sum(axis=2)                  # [wg, 2 physics waves]
            wait = R[:, 3, 1:3]
            cwait = R[:, 3, 0]
            tot = busy + wait
            wg = tot.max(axis=1)
            print("t %2d per-wave: physics busy median %.0f p90 %.0f max %.0f | barrier wait median %.0f max %.0f | controller wait median %.0f | busy+wait per workgroup median %.0f max %.0f | "
                  "max over workgroups of max-wave busy %.0f, of sum-of-max estimate n/a" % (t, np.median(busy), np.percentile(busy, 90), busy.max(), np.median(wait), wait.max(), np.median(cwait),
                                                                                     np.median(wg), wg.max(), busy.max(axis=1).max()), flush=True)
            i = int(np.argmax(wg))
            print("      slowest workgroup %d: wave busy %s wait %s; phases of its waves (arm, statics, bb+rod, tree, generic): %s" % (i, busy[i].round(0), wait[i].round(0), R[i, 1:3][:, [0, 1, 2, 8, 9]].round(0).tolist()), flush=True)
            continue                                       # ticks -> microseconds per env step; the other slots are counts (first active lane of the wave)
        for lab, v in (("median", np.median(Wf, axis=0)), ("p90", np.percentile(Wf, 90, axis=0)), ("max", Wf.max(axis=0))):
            print("t %2d %6s per workgroup: " % (t, lab) + "  ".join("%s %.0f" % (names[i], v[i]) for i in range(0, 10)), flush=True)
        Cf = CN.astype(np.float64)
        for lab, v in (("median", np.median(Cf, axis=0)), ("max", Cf.max(axis=0))):
            print("t %2d %6s wave-level counts per env step: " % (t, lab) + "  ".join("%s %.0f" % (cnames[i], v[i]) for i in range(8)), flush=True)
