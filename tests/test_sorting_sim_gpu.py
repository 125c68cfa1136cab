"""Sorting_Sim mirror end to end on the GPU (protocol + metric plumbing; the physics parity lives in test_gpu_parity_sorting.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def test_sorting_sim_protocol_and_metrics():
    from d3il_amd.simulation.metrics import sorting_metrics
    from d3il_amd.simulation.sorting_sim import Sorting_Sim, completion_order_codes
    from tests.dist_sim_worker import PushToBinAgent
    sim = Sorting_Sim(seed=0, device="cuda:0", render=False, n_cores=1, n_contexts=10, n_trajectories_per_context=3, max_steps_per_episode=140)
    assert sim.mode_keys.tolist() == completion_order_codes(4) == [48, 80, 96, 144, 160, 192] and sim.n_mode == 6
    res = sim.test_agent(PushToBinAgent())
    assert set(res) == {"score", "Metrics/successes", "Metrics/KL", "Metrics/entropy"}
    r = sim.last_rollout
    mode, success = r["mode"].cpu().numpy().reshape(10, 3), r["success"].cpu().numpy().reshape(10, 3)
    # the rollouts of a context are identical (deterministic policy, same context): their results agree
    assert (mode == mode[:, :1]).all() and not success.any()
    # the scripted push delivers red_1 in some contexts (a third of them at this horizon), nothing else yet
    # (a stray blue box may reach its bin first: any code with one completed entry is legitimate)
    assert set(np.unique(mode)) <= {240, 112, 176} and (mode == 112).sum() >= 3, np.unique(mode, return_counts=True)
    assert not bool((r["flags"] & ((1 << 16) | (1 << 18))).any())
    assert r["counts"][-1] == 0 and res["Metrics/successes"] == 0.0
    # metric tail from the integer tables equals the direct formula on a synthetic outcome table
    counts = np.zeros((10, 6), dtype=np.int64)
    counts[np.arange(10), np.arange(10) % 6] = 2
    sr, ent, kl, score = sorting_metrics(counts, 20, 30, 3, sim.mode_encoding.numpy())
    assert abs(sr - 20 / 30) < 1e-7 and abs(ent) < 1e-6 and abs(score - (sr - kl)) < 1e-7


def test_sorting_env_protocol():
    from d3il_amd.envs.sorting import SortingVecEnv, sample_contexts
    env = SortingVecEnv(33, device=0)
    q, iters, err = env.start()
    assert err < 1e-3
    with pytest.raises(ValueError):
        env.reset()                                                   # no context yet
    with pytest.raises(ValueError):
        env.reset(context=np.zeros((33, 14)))                         # Pushing-shaped context
    ctx = sample_contexts(33, 4, seed=2)
    obs = env.reset(context=ctx)
    assert obs.shape == (33, 14) and obs.dtype == torch.float32
    np.testing.assert_allclose(obs[:, [2, 3, 5, 6, 8, 9, 11, 12]].cpu().numpy(), ctx.reshape(33, 4, 7)[:, :, :2].reshape(33, 8), atol=1e-4)   # one sub-step after placement: overlapping neighbours have started to separate
    pos, quat = env.box_state()
    assert pos.shape == (33, 4, 3) and quat.shape == (33, 4, 4)
    a = torch.cat([env.robot_state()[:, :3], torch.tensor([0.0, 1, 0, 0], dtype=torch.float64, device=env.device).expand(33, 4)], 1).contiguous()
    obs2, rew, done, info = env.step(a)
    assert float(rew.abs().max()) == 0 and not bool(done.any()) and (info["mode"] == 240).all() and not bool(info["success"].any())
    # masked reset: only the selected environments restart
    for _ in range(3):
        env.step(a)
    mask = torch.zeros(33, dtype=torch.uint8, device=env.device); mask[5] = 1
    sc0 = env.step_count[:33].clone()
    env.reset(mask=mask)
    torch.cuda.synchronize()
    sc1 = env.step_count[:33]
    assert int(sc1[5]) == 0 and torch.equal(sc1[torch.arange(33) != 5], sc0[torch.arange(33) != 5])
    env.close()
