"""Pushing at scale with a scripted two-phase policy (red cube -> red target, then green cube -> green target): success / mode
bookkeeping, solver health and physical sanity over full 400-step episodes."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from d3il_amd.envs.pushing import BlockPushVecEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 960
ctx60 = np.load(os.path.join(ROOT, "d3il_amd", "data", "pushing_test_contexts.npy"))
ctx = ctx60[np.arange(n) % 60]
iq = np.load(os.path.join(ROOT, "tests", "golden", "ref_offline_ik.npz"))["avoiding__traj_last"]
env = BlockPushVecEnv(n, device=0)
env.set_init_qpos(iq)
env.reset(context=ctx)
dev = env.device
des = env.robot_state()[:, :2].clone(); z = env.robot_state()[:, 2:3].clone()
quat = torch.tensor([0.0, 1, 0, 0], dtype=torch.float64, device=dev).expand(n, 4)
goals = torch.tensor([[0.42, 0.3], [0.63, 0.3]], dtype=torch.float64, device=dev)
phase = torch.zeros(n, dtype=torch.long, device=dev)         # 0: red cube, 1: green cube
finished = torch.zeros(n, dtype=torch.bool, device=dev)
succ = torch.zeros(n, dtype=torch.bool, device=dev); mode = torch.full((n,), -9, dtype=torch.long, device=dev)
env.set_timing(True); ms = []
for t in range(400):
    o = env.obs.to(torch.float64)
    box = torch.where(phase.unsqueeze(1) == 0, o[:, 2:4], o[:, 5:7])
    goal = goals[phase]
    togo = goal - box
    dist = togo.norm(dim=1, keepdim=True)
    phase = torch.where((dist.squeeze(1) < 0.03) & (phase == 0), torch.ones_like(phase), phase)
    dirn = togo / dist.clamp_min(1e-9)
    behind = box - dirn * 0.055                                # stand-off point behind the cube, on the line to its goal
    off = des - behind
    lateral = off - (off * dirn).sum(1, keepdim=True) * dirn
    aligned = (lateral.norm(dim=1, keepdim=True) < 0.012) & ((off * dirn).sum(1, keepdim=True) < 0.02)
    target = torch.where(aligned, box + dirn * 0.0, behind)
    # go around the cube when the straight line to the stand-off point crosses it
    d = target - des
    step = d / d.norm(dim=1, keepdim=True).clamp_min(1e-9) * torch.minimum(d.norm(dim=1, keepdim=True), torch.full_like(dist, 0.006))
    near = ((des - box).norm(dim=1, keepdim=True) < 0.06) & ~aligned
    away = (des - box) / (des - box).norm(dim=1, keepdim=True).clamp_min(1e-9)
    tang = torch.stack((-away[:, 1], away[:, 0]), dim=1)
    tang = tang * torch.sign((tang * (behind - des)).sum(1, keepdim=True) + 1e-12)
    step = torch.where(near, 0.006 * (0.6 * tang + 0.4 * away), step)
    if os.environ.get("PUSH_POLICY") == "random":     # random walk with a drift towards the current cube: frequent glancing contacts
        step = 0.5 * step + (torch.rand(n, 2, dtype=torch.float64, device=dev) * 0.02 - 0.01)
    des = des + step
    act = torch.cat([des, z, quat], dim=1).contiguous()
    if os.environ.get("PUSH_TRACE_FAIL"):
        torch.cuda.synchronize(); st0, fl0, sc0 = env.get_state()
    obs, rew, done, info = env.step(act)
    if os.environ.get("PUSH_TRACE_FAIL"):
        torch.cuda.synchronize(); st1, fl1, sc1 = env.get_state()
        bad = np.nonzero(((fl1 >> 16) & 1) & ~((fl0 >> 16) & 1))[0]
        if len(bad):
            e = int(bad[0])
            np.savez(os.path.join(ROOT, "gpurun_out", "push_fail_%d.npz" % t), state=st0[:, e], flags=fl0[e], step=sc0[e], action=act[e].cpu().numpy(), state1=st1[:, e], flags1=fl1[e])
            print("step", t, "new failures", len(bad), "first env", e, flush=True)
    newly = ~finished & done.bool()
    succ = torch.where(newly, info["success"].bool(), succ); mode = torch.where(newly, info["mode"].long(), mode)
    finished |= done.bool()
    if t % 16 == 15:
        torch.cuda.synchronize(); ms.append(env.last_step_ms())
torch.cuda.synchronize()
st, fl, sc = env.get_state()
pos, quatb = env.box_state()
print("kernel ms mean %.2f max %.2f" % (np.mean(ms), np.max(ms)))
print("success rate %.3f  modes %s" % (float(succ.float().mean()), np.unique(mode.cpu().numpy(), return_counts=True)))
print("flags: fail %d overflow %d offtable %d ; finite %s" % (np.sum((fl >> 16) & 1), np.sum((fl >> 18) & 1), np.sum((fl >> 19) & 1), np.isfinite(st).all()))
print("cube z range", float(pos[:, :, 2].min()), float(pos[:, :, 2].max()), " mean_distance", float(info["mean_distance"].mean()))
