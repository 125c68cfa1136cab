"""Resting-regime step time of the generic engine's tasks (Sorting-4, Inserting): every rod holds its start pose, the cubes lie where the context put
them; optionally (--wall, Inserting) every rod is first driven into a wall of the left gate and keeps pressing (the arm-alone island with rod <-> wall
contacts).  python tools/gpu_gen_rest_time.py [--envs 4096]"""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(task, n, wall=False, clear=False):
    if task == "sorting":
        from d3il_amd.envs.sorting import SortingVecEnv, sample_contexts
        env = SortingVecEnv(n, device=0, max_steps_per_episode=100000); ctx = sample_contexts(60, 4, seed=0)
    else:
        from d3il_amd.envs.inserting import GateInsertionVecEnv, sample_contexts
        env = GateInsertionVecEnv(n, device=0, max_steps_per_episode=100000); ctx = sample_contexts(60, seed=0)
        if clear:      # the cubes out of the rod's way (right half of the table): the arm is alone with its wall contact
            ctx = ctx.reshape(60, 3, 7).copy(); ctx[:, :, 0] = [0.66, 0.72, 0.64]; ctx[:, :, 1] = [-0.2, -0.08, 0.04]; ctx = ctx.reshape(60, 21)
    env.start()
    env.reset(context=ctx[np.arange(n) % 60])
    z = env.robot_state()[:, 2:3].clone(); des = env.obs[:, :2].to(torch.float64).clone()
    quat = torch.tensor([0.0, 1, 0, 0], dtype=torch.float64, device=des.device).expand(n, 4)
    def act(): return torch.cat([des, z, quat], 1).contiguous()
    for t in range(30): env.step(act())
    if wall:
        for t in range(110):
            tgt = torch.tensor([0.45, 0.10] if t < 70 else [0.40, 0.26], dtype=torch.float64, device=des.device).expand(n, 2)
            d = tgt - des; nn = d.norm(dim=1, keepdim=True)
            des = des + d / nn.clamp_min(1e-9) * torch.minimum(nn, torch.full_like(nn, 0.006))
            env.step(act())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in range(50): env.step(act())
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    fl = env.flags[:n].cpu().numpy()
    print("%s%s%s: %.3f ms per step at %d envs (%.3f M env-steps/s), flagged %d" % (task, " rod on wall" if wall else " at rest", " (cubes out of the way)" if clear else "", dt * 1e3, n, n / dt / 1e6, int(((fl >> 16) & 0xD).astype(bool).sum())))
    env.close()


n = int(sys.argv[sys.argv.index("--envs") + 1]) if "--envs" in sys.argv else 4096
run("sorting", n); run("inserting", n); run("inserting", n, wall=True); run("inserting", n, wall=True, clear=True)
