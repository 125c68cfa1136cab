"""Worker for the multi-process GPU test: one rank of a torch.distributed job running the batched Sim classes.

Launched by tests/test_gpu_multiprocess.py through `python -m torch.distributed.run`; every rank uses cuda:0 (the GPU box has
one GPU) with the gloo backend, which exercises the same sharding / all-reduce code path as one-rank-per-GPU with RCCL."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from d3il_amd import distributed as D  # noqa: E402
from d3il_amd.simulation.avoiding_sim import Avoiding_Sim  # noqa: E402
from d3il_amd.simulation.pushing_sim import Pushing_Sim  # noqa: E402
from d3il_amd.simulation.sorting_sim import Sorting_Sim  # noqa: E402
from d3il_amd.simulation.stacking_sim import Stacking_Sim  # noqa: E402


class ChaseAgent:
    """deterministic batched policy for Pushing: walk towards the red cube"""

    def reset(self):
        pass

    def predict_batch(self, obs10):
        d = obs10[:, 4:6] - obs10[:, 0:2]
        n = d.norm(dim=1, keepdim=True).clamp_min(1e-9)
        return d / n * torch.minimum(n, torch.full_like(n, 0.006))


class PushToBinAgent:
    """deterministic batched policy for Sorting: get behind red_1, then push it towards the red bin"""

    def reset(self):
        self.t = 0

    def predict_batch(self, obs16):
        des, box = obs16[:, 0:2], obs16[:, 4:6]
        aligned = ((des[:, 0] - box[:, 0]).abs() < 0.008) & (des[:, 1] < box[:, 1] - 0.02)
        target = torch.where(aligned[:, None], torch.stack([box[:, 0], torch.full_like(box[:, 0], 0.36)], 1), box + torch.tensor([0.0, -0.06], dtype=des.dtype, device=des.device))
        d = target - des
        n = d.norm(dim=1, keepdim=True)
        return d / n.clamp_min(1e-9) * torch.minimum(n, torch.full_like(n, 0.006))


class WiggleAgent:
    """deterministic batched policy for Avoiding: forward drift with an obs-dependent lateral wiggle"""

    def reset(self):
        pass

    def predict_batch(self, obs4):
        x, y = obs4[:, 2], obs4[:, 3]
        return torch.stack((0.006 * torch.sin(40.0 * y), torch.full_like(y, 0.004)), dim=1)


def main():
    rank, world = D.init_from_env("gloo")
    out = {}
    sim = Pushing_Sim(seed=0, device="cuda:0", render=False, n_cores=1, n_contexts=9, n_trajectories_per_context=2, max_steps_per_episode=40)
    succ, mode, dist = sim.test_agent(ChaseAgent())
    r = sim.last_rollout
    out["pushing"] = dict(counts=[int(v) for v in r["counts"]], success_rate=r["success_rate"], entropy=r["entropy"],
                          mean_distance=r["mean_distance_all"], shard=list(r["shard"]))
    sim2 = Avoiding_Sim(seed=0, device="cuda:0", render=False, n_cores=1, n_trajectories=37, max_steps_per_episode=60)
    successes, entropy = sim2.test_agent(WiggleAgent())
    r2 = sim2.last_rollout
    out["avoiding"] = dict(counts=[int(v) for v in r2["counts"]], entropy=float(entropy), shard=list(r2["shard"]))
    sim3 = Sorting_Sim(seed=0, device="cuda:0", render=False, n_cores=1, n_contexts=11, n_trajectories_per_context=2, max_steps_per_episode=130)
    res3 = sim3.test_agent(PushToBinAgent())
    r3 = sim3.last_rollout
    out["sorting"] = dict(counts=[int(v) for v in r3["counts"]], mode_hist=[int(v) for v in r3["mode_hist"]], shard=list(r3["shard"]),
                          successes=res3["Metrics/successes"], entropy=res3["Metrics/entropy"])
    # Stacking (BASELINE config 5's Sim class): scripted pick-and-place of the first box on 5 contexts x 2 rollouts, sharded over the ranks
    import numpy as np
    from d3il_amd.agents import ScriptedStackPolicy
    from d3il_amd.controllers.scripted_stacking import build_trajectory
    from d3il_amd.distributed import shard_range
    from d3il_amd.envs.stacking import load_test_contexts
    from d3il_amd.model import blob
    q0 = np.load(os.path.join(ROOT, "tests", "golden", "ref_offline_ik.npz"))["stacking__traj_last"]
    nctx, ntraj = 5, 2
    ctx = load_test_contexts()[:nctx]
    tables = [build_trajectory(blob.load_json("stacking"), q0, c, n_boxes=1, speed=0.8) for c in ctx]
    lo, hi = shard_range(nctx * ntraj, rank, world)
    sim4 = Stacking_Sim(seed=0, device="cuda:0", render=False, n_cores=1, n_contexts=nctx, n_trajectories_per_context=ntraj,
                        max_steps_per_episode=max(len(t) for t in tables) + 5, contexts=ctx)
    sim4.test_agent(ScriptedStackPolicy(tables, torch.arange(lo, hi) // ntraj, device="cuda:0"))
    r4 = sim4.last_rollout
    out["stacking"] = dict(counts=[int(v) for v in r4["counts"]], shard=list(r4["shard"]), successes_1_box=r4["metrics"]["successes_1_box"])
    # Aligning: scripted inside / outside pushes on 7 contexts x 2 rollouts; the policy's per-lane behaviour follows the GLOBAL rollout index
    from d3il_amd.agents import ScriptedAlignPolicy
    from d3il_amd.simulation.aligning_sim import Aligning_Sim
    lo5, hi5 = shard_range(14, rank, world)
    sim5 = Aligning_Sim(seed=0, device="cuda:0", render=False, n_cores=1, n_contexts=7, n_trajectories_per_context=2, max_steps_per_episode=110)
    sim5.test_agent(ScriptedAlignPolicy(inside=(torch.arange(lo5, hi5) % 2 == 0), device="cuda:0"))
    r5 = sim5.last_rollout
    out["aligning"] = dict(counts=[int(v) for v in r5["counts"]], shard=list(r5["shard"]), mean_distance=r5["mean_distance_all"],
                           modes=sorted(set(int(m) for m in r5["mode"].cpu().tolist())))
    if rank == 0:
        print("RESULT " + json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
