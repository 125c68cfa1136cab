"""Pushing_Sim mirror end to end on the GPU (protocol + metric plumbing; the physics parity lives in test_gpu_parity_pushing.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


class _ChaseAgent:
    """batched scripted policy: push the red cube towards its (pairing 0) target"""

    def reset(self):
        pass

    def predict_batch(self, obs10):
        des, box = obs10[:, 0:2], obs10[:, 4:6]
        goal = torch.tensor([0.42, 0.3], dtype=obs10.dtype, device=obs10.device)
        away = box - goal
        behind = box + away / away.norm(dim=1, keepdim=True).clamp_min(1e-9) * 0.06     # a point behind the cube
        far = (des - behind).norm(dim=1, keepdim=True) > 0.02
        target = torch.where(far & ((des - box).norm(dim=1, keepdim=True) > 0.05), behind, goal.expand_as(des))
        d = target - des
        return d / d.norm(dim=1, keepdim=True).clamp_min(1e-9) * 0.008


def test_pushing_sim_protocol_and_metrics():
    from d3il_amd.simulation.pushing_sim import Pushing_Sim
    sim = Pushing_Sim(seed=0, device="cuda:0", render=False, n_cores=1, n_contexts=12, n_trajectories_per_context=4, max_steps_per_episode=60)
    successes, modes, dist = sim.test_agent(_ChaseAgent())
    assert successes.shape == (12, 4) and modes.shape == (12, 4) and dist.shape == (12, 4)
    r = sim.last_rollout
    # the 4 rollouts of a context are identical (deterministic policy): their results agree
    assert torch.equal(modes, modes[:, :1].expand_as(modes)) and torch.equal(dist, dist[:, :1].expand_as(dist))
    assert 0.0 <= r["success_rate"] <= 1.0 and 0.0 <= r["entropy"] <= 1.0 + 1e-6
    assert float(dist.max()) < 0.5 and float(dist.min()) > 0.0
    assert not bool((r["flags"] & ((1 << 16) | (1 << 18))).any())
    # metric tail from the integer tables equals the direct formula
    su, mo = successes.cpu().numpy(), modes.cpu().numpy()
    counts = np.array([[np.sum((mo[c] == m) & (su[c] == 1)) for m in range(4)] for c in range(12)])
    assert np.array_equal(counts.reshape(-1), r["counts"][:-1]) and int(su.sum()) == r["counts"][-1]


def test_stand_in_policy_is_deterministic_and_bounded():
    from d3il_amd.agents import RandomResidualMLPPolicy
    p1, p2 = RandomResidualMLPPolicy(device="cuda:0"), RandomResidualMLPPolicy(device="cuda:0")
    x = torch.randn(64, 10, device="cuda:0", dtype=torch.float64)
    a, b = p1.predict_batch(x), p2.predict_batch(x)
    assert torch.equal(a, b) and a.shape == (64, 2) and float(a.abs().max()) <= 0.01 + 1e-9
    assert sum(p.numel() for p in p1.parameters()) == 10 * 128 + 128 + 6 * (128 * 128 + 128) + 128 * 2 + 2
