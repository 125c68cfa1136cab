"""Batched policy adapter: row i of predict_batch == the reference protocol's predict(obs[i]) (bc_agent.py:240-271)."""
import numpy as np
import torch

from d3il_amd.agents import BatchedBCAgent, RowwiseAgent


class _Scaler:
    """Same arithmetic as agents/utils/scaler.py:72-113 (scale_data=True)."""

    def __init__(self, obs_dim, act_dim):
        g = torch.Generator().manual_seed(0)
        self.x_mean, self.x_std = torch.randn(obs_dim, generator=g), torch.rand(obs_dim, generator=g) + 0.5
        self.y_mean, self.y_std = torch.randn(act_dim, generator=g) * 0.01, torch.rand(act_dim, generator=g) * 0.01

    def scale_input(self, x):
        return ((x - self.x_mean) / (self.x_std + 1e-12 * torch.ones(self.x_std.shape))).to(torch.float32)

    def inverse_scale_output(self, y):
        return y * (self.y_std + 1e-12 * torch.ones(self.y_std.shape)) + self.y_mean


class _RefLikeBCAgent:
    """Stand-in with the attribute names and predict() body of the reference BC_Agent (batch 1, numpy in/out)."""

    def __init__(self, obs_dim=4, act_dim=2):
        torch.manual_seed(0)
        self.model = torch.nn.Sequential(torch.nn.Linear(obs_dim, 64), torch.nn.Mish(), torch.nn.Linear(64, 64), torch.nn.Mish(),
                                         torch.nn.Linear(64, act_dim))
        self.scaler = _Scaler(obs_dim, act_dim)
        self.min_action, self.max_action = -1.0, 1.0
        self.device = "cpu"

    def reset(self):
        pass

    @torch.no_grad()
    def predict(self, state):
        self.model.eval()
        state = torch.from_numpy(state).float().to(self.device).unsqueeze(0).unsqueeze(0)
        state = self.scaler.scale_input(state)
        out = self.model(state)
        out = out.clamp_(self.min_action, self.max_action)
        return self.scaler.inverse_scale_output(out).detach().cpu().numpy()[0]


def test_batched_bc_adapter_matches_reference_protocol():
    agent = _RefLikeBCAgent()
    obs = torch.randn(37, 4, dtype=torch.float64) * 0.3
    batched = BatchedBCAgent(agent).predict_batch(obs)
    assert batched.shape == (37, 2)
    for i in range(37):
        ref = agent.predict(obs[i].numpy())          # [1, 2]
        np.testing.assert_allclose(batched[i].numpy(), ref[0], rtol=0, atol=1e-7)
    rowwise = RowwiseAgent(agent).predict_batch(obs)
    np.testing.assert_allclose(rowwise.numpy(), batched.numpy().astype(np.float64), atol=1e-7)
