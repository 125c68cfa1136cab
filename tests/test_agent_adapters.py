"""Agent adapters of the batched sims (d3il_amd/agents.py).

* ``RowwiseAgent`` - the fallback for reference agents without ``predict_batch`` - must keep ONE AGENT STATE PER LANE:
  most reference agents hold an observation deque and counters inside ``predict`` (ddpm_agent.py:223-227, beso_agent.py:354-355), so a single
  instance stepped through N lock-stepped environments would mix their histories.  The test drives a window-3 agent through the
  adapter and through N sequential single-environment rollouts and requires identical actions.
* ``BatchedBCAgent`` wraps a reference ``BC_Agent`` object (attributes ``model``, ``scaler``, ``min_action``, ``max_action``); the native
  batched policies and their pinning against the reference's own agents live in d3il_amd/policies.py / tests/test_policies.py.
"""
from collections import deque

import numpy as np
import torch

from d3il_amd.agents import BatchedBCAgent, RowwiseAgent, as_batched


class _WindowAgent:
    """Reference-style stateful agent: predict() appends to a deque(maxlen=3) and answers from the whole window; a step counter counts calls."""

    def __init__(self):
        torch.manual_seed(0)
        self.model = torch.nn.Linear(3 * 4, 2)
        self.obs_context = deque(maxlen=3)
        self.n_calls = 0

    def reset(self):
        self.obs_context.clear()
        self.n_calls = 0

    @torch.no_grad()
    def predict(self, state):
        self.obs_context.append(torch.from_numpy(state).float())
        self.n_calls += 1
        win = list(self.obs_context) + [torch.zeros(4)] * (3 - len(self.obs_context))
        return (self.model(torch.cat(win)) * (1 + 0.01 * self.n_calls)).numpy()[None]


def test_rowwise_fallback_keeps_one_history_per_lane():
    n, T = 6, 7
    obs = np.random.default_rng(0).normal(size=(T, n, 4))
    agent = _WindowAgent()
    batched = as_batched(agent, n)
    assert isinstance(batched, RowwiseAgent) and len(batched.lanes) == n
    assert all(l.model is agent.model for l in batched.lanes)                 # the network is shared, the history is not
    assert len({id(l.obs_context) for l in batched.lanes}) == n
    batched.reset()
    got = np.stack([batched.predict_batch(torch.as_tensor(obs[t])).numpy() for t in range(T)])
    for e in range(n):                                                           # sequential single-environment rollouts
        ref_agent = _WindowAgent()
        ref_agent.reset()
        for t in range(T):
            np.testing.assert_allclose(got[t, e], ref_agent.predict(obs[t, e])[0], atol=1e-6)
    # a lane that starts its next trajectory gets a fresh history, the others keep theirs
    mask = torch.zeros(n, dtype=torch.uint8); mask[3] = 1
    batched.begin_episodes(mask)
    assert len(batched.lanes[3].obs_context) == 0 and batched.lanes[3].n_calls == 0 and len(batched.lanes[2].obs_context) == 3


class _Scaler:
    def __init__(self):
        self.x_mean, self.x_std = torch.tensor([0.1, -0.2, 0.3, 0.0]), torch.tensor([0.5, 0.7, 0.9, 1.1])
        self.y_mean, self.y_std = torch.tensor([0.001, -0.002]), torch.tensor([0.004, 0.006])

    def scale_input(self, x):
        return ((x - self.x_mean) / (self.x_std + 1e-12)).to(torch.float32)

    def inverse_scale_output(self, y):
        return y * (self.y_std + 1e-12) + self.y_mean


def test_batched_bc_agent_wrapper_uses_the_wrapped_objects():
    class A:
        pass
    a = A()
    torch.manual_seed(1)
    a.model = torch.nn.Sequential(torch.nn.Linear(4, 16), torch.nn.Mish(), torch.nn.Linear(16, 2))
    a.scaler, a.min_action, a.max_action = _Scaler(), -1.0, 1.0
    obs = torch.randn(9, 4, dtype=torch.float64)
    out = BatchedBCAgent(a).predict_batch(obs)
    ref = a.scaler.inverse_scale_output(a.model(a.scaler.scale_input(obs.float())).clamp(-1, 1))
    np.testing.assert_allclose(out.detach().numpy(), ref.detach().numpy(), atol=1e-7)
