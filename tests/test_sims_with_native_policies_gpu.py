"""BASELINE configs 4 / 5 through the Sim mirrors with the native batched policies (d3il_amd/policies.py; fixed random weights - there
are no checkpoints offline): the plumbing a D3IL user gets when the Hydra `simulation._target_` points at this package and the agent
offers `predict_batch` (INTEGRATION.md sections 1 and 7).  Physics parity lives in the test_gpu_parity_* files."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def test_sorting_sim_with_ddpm_policy():
    import bench
    from d3il_amd.simulation.sorting_sim import Sorting_Sim
    sim = Sorting_Sim(seed=0, device="cuda:0", render=False, n_cores=1, n_contexts=6, n_trajectories_per_context=4, max_steps_per_episode=25)
    pol = bench._random_ddpm(16, torch.device("cuda:0"))           # obs 16 = desired xy + the 14-d observation (configs/sorting_4_config.yaml:47)
    res = sim.test_agent(pol)
    assert set(res) == {"score", "Metrics/successes", "Metrics/KL", "Metrics/entropy"}     # KL / entropy of an empty success table follow the reference
    r = sim.last_rollout
    assert r["mode"].shape[0] == 24 and not bool((r["flags"] & ((1 << 16) | (1 << 18))).any())     # no solver failure, no contact overflow
    assert res["Metrics/successes"] == 0.0                           # nothing is sorted within 25 steps


def test_stacking_sim_with_beso_policy():
    import bench
    from d3il_amd.simulation.stacking_sim import Stacking_Sim
    sim = Stacking_Sim(seed=0, device="cuda:0", render=False, n_cores=1, n_contexts=5, n_trajectories_per_context=3, max_steps_per_episode=12)
    pol = bench._random_beso(torch.device("cuda:0"))
    pol.use_graph = False
    res = sim.test_agent(pol)
    successes, mode_encoding = res                                    # the reference's return value: two [n_contexts, n_trajectories] tables
    assert tuple(successes.shape) == (5, 3) == tuple(mode_encoding.shape) and float(successes.sum()) == 0.0
    assert float(sim.last_rollout["metrics"]["successes"]) == 0.0
    r = sim.last_rollout
    assert r["mode"].shape[0] == 15 and not bool((r["flags"] & ((1 << 16) | (1 << 18))).any())
    assert not bool(r["success"].any()) and pol.obs_hist.len.tolist() == [5] * 15      # window 5 filled, per-lane histories in lock step
