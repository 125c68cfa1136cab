"""Native batched policies (d3il_amd/policies.py) against the reference's own agents: tests/golden/ref_agents.npz holds fixed-seed
weights, scaler statistics, observations, the sampler noise and the outputs of BC_Agent / DiffusionAgent / BesoAgent.predict rolled
out batch-1 per environment (tests/golden/gen_agent_goldens.py, run where the reference is).  Row i of predict_batch must equal the
reference's predict for environment i at every step - including the history windows (BESO: 5 observations, 4 previous actions) and
the single EMA swap.  Runs on the CPU; the gpu-marked variant runs the same replay on cuda:0."""
import os

import numpy as np
import pytest
import torch

from d3il_amd import policies as P

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_agents.npz"))


def _sd(prefix):
    return {k[len(prefix):].replace("__", "."): torch.as_tensor(G[k]) for k in G.files if k.startswith(prefix)}


def _scaler(tag, dev):
    return P.Scaler(G[tag + "_x_mean"], G[tag + "_x_std"], G[tag + "_y_mean"], G[tag + "_y_std"], G[tag + "_y_bounds"], device=dev)


class _Bank:
    def __init__(self, bank, dev):
        self.bank, self.call, self.dev = torch.as_tensor(bank), 0, dev

    def __call__(self, shape):
        b = self.bank[self.call]
        self.call += 1
        out = b[:, 0, :shape[1]] if len(shape) == 2 else b[:, :shape[1], :shape[2]]
        return out.to(self.dev).contiguous()


def _bc(dev):
    model = P.ResidualMLP(10, 32, 4, 2).to(dev)
    model.load_state_dict(_sd("bc_sd__"))
    sc = _scaler("bc", dev)
    pol = P.BCPolicy(model, sc, sc.y_bounds[0], sc.y_bounds[1])
    obs, ref = G["bc_obs"], G["bc_ref"]
    worst = 0.0
    for t in range(obs.shape[1]):
        a = pol.predict_batch(torch.as_tensor(obs[:, t], device=dev))
        worst = max(worst, float(np.abs(a.cpu().numpy() - ref[:, t]).max()))
    return worst


def _ddpm(dev):
    model = P.DiffusionMLP(2, 16, 8, 32, 4).to(dev)
    bank = _Bank(G["ddpm_noise"], dev)
    pol = P.DDPMPolicy(model, _scaler("ddpm", dev), n_timesteps=4, noise_fn=bank)
    pol.load_reference_state_dict(_sd("ddpm_sd__"))
    pol.use_ema([G[k] for k in sorted(k for k in G.files if k.startswith("ddpm_ema__"))])
    obs, ref = G["ddpm_obs"], G["ddpm_ref"]
    worst = 0.0
    for t in range(obs.shape[1]):
        a = pol.predict_batch(torch.as_tensor(obs[:, t], device=dev))
        worst = max(worst, float(np.abs(a.cpu().numpy() - ref[:, t]).max()))
    return worst


def _beso(dev, staggered=False):
    inner = P.DiffusionGPT(20, 8, 32, 2, 4, 5, linear_output=True).to(dev)
    bank = _Bank(G["beso_noise"], dev)
    pol = P.BESOPolicy(inner, _scaler("beso", dev), window_size=5, num_sampling_steps=16, sigma_min=0.01, sigma_max=1.0, sigma_data=0.5, noise_fn=bank)
    pol.load_reference_state_dict(_sd("beso_sd__"))
    obs, ref = G["beso_obs"], G["beso_ref"]
    worst = 0.0
    for t in range(obs.shape[1]):
        a = pol.predict_batch(torch.as_tensor(obs[:, t], device=dev))
        worst = max(worst, float(np.abs(a.cpu().numpy() - ref[:, t]).max()))
    return worst


def test_bc_policy_rows_equal_reference_predict():
    assert _bc("cpu") < 2e-6


def test_ddpm_policy_rows_equal_reference_predict():
    assert _ddpm("cpu") < 2e-6


def test_beso_policy_rows_equal_reference_predict():
    assert _beso("cpu") < 5e-6


def test_history_restart_of_single_lanes():
    """begin_episodes(mask): a lane that restarts its trajectory gets a fresh window while the others keep theirs - the restarted lane
    must reproduce the reference's first steps again (the reference agent.reset() clears its deques)."""
    dev = "cpu"
    inner = P.DiffusionGPT(20, 8, 32, 2, 4, 5, linear_output=True)
    pol = P.BESOPolicy(inner, _scaler("beso", dev), window_size=5, num_sampling_steps=16, sigma_min=0.01, sigma_max=1.0, sigma_data=0.5,
                       noise_fn=lambda shape: torch.zeros(shape))
    pol.load_reference_state_dict(_sd("beso_sd__"))
    obs = torch.as_tensor(G["beso_obs"])
    first = [pol.predict_batch(obs[:, t]).clone() for t in range(3)]
    mask = torch.zeros(obs.shape[0], dtype=torch.uint8); mask[2] = 1
    pol.begin_episodes(mask)
    again = [pol.predict_batch(torch.where(mask.bool().unsqueeze(1), obs[:, t], obs[:, 3 + t])) for t in range(3)]
    for t in range(3):      # lane 2 sees its first observations again with an empty history: same outputs as at the start
        np.testing.assert_allclose(again[t][2].numpy(), first[t][2].numpy(), atol=1e-6)
    assert pol.obs_hist.len.tolist() == [5, 5, 3, 5, 5]


@pytest.mark.gpu
def test_policies_on_the_gpu():
    assert torch.cuda.is_available()
    assert _bc("cuda:0") < 1e-5 and _ddpm("cuda:0") < 1e-5 and _beso("cuda:0") < 2e-5
