"""Batched adapters for the reference's policy protocol (the consumer side of the rollout boundary).

The reference agents are batch-1 and numpy-in/numpy-out: ``agent.predict(np.ndarray[obs]) -> np.ndarray[1, act]``
(agents/base_agent.py:110-122), with a host<->device round trip inside every call (agents/bc_agent.py:259-271).  The
adapters below run the *same* computation once per step on the whole environment batch, on device-resident
observations, and are what ``Avoiding_Sim.test_agent`` looks for (``predict_batch``).  Networks and scalers are the
reference's own objects - nothing is re-implemented here.
"""
from __future__ import annotations

import torch


class BatchedBCAgent:
    """Wraps a reference ``BC_Agent`` (agents/bc_agent.py): attributes used are ``model``, ``scaler``
    (``scale_input`` / ``inverse_scale_output``, agents/utils/scaler.py:72-113), ``min_action``, ``max_action``."""

    def __init__(self, agent):
        self.agent = agent

    def reset(self):
        if hasattr(self.agent, "reset"):
            self.agent.reset()

    @torch.no_grad()
    def predict_batch(self, obs: torch.Tensor) -> torch.Tensor:
        """obs [N, obs_dim] on the policy's device -> actions [N, act_dim]; row i equals ``agent.predict(obs[i])[0]``
        (bc_agent.py:240-271 with the batch dimension N instead of 1)."""
        a = self.agent
        a.model.eval()
        state = obs.to(torch.float32).unsqueeze(1)            # [N, 1, obs]; the reference builds [1, 1, obs]
        state = a.scaler.scale_input(state)
        out = a.model(state)
        out = out.clamp_(a.min_action, a.max_action)
        pred = a.scaler.inverse_scale_output(out)
        return pred[:, 0]

    def predict(self, state, *args, **kwargs):               # the reference protocol still works
        return self.agent.predict(state, *args, **kwargs)


class RowwiseAgent:
    """Fallback adapter for any reference agent: calls ``predict`` row by row (host round trip per env, slow)."""

    def __init__(self, agent):
        self.agent = agent

    def reset(self):
        if hasattr(self.agent, "reset"):
            self.agent.reset()

    def predict_batch(self, obs: torch.Tensor) -> torch.Tensor:
        import numpy as np
        rows = obs.detach().cpu().numpy()
        acts = np.stack([np.asarray(self.agent.predict(r)).reshape(-1) for r in rows])
        return torch.as_tensor(acts, dtype=torch.float64, device=obs.device)


class RandomResidualMLPPolicy(torch.nn.Module):
    """Stand-in policy of BASELINE config 3: the architecture of the reference's BC policy for Pushing
    (agents/models/common/mlp.py:114-182 as configured by configs/agents/bc_agent.yaml:11-22 and
    configs/pushing_config.yaml: input 10 = desired xy + obs 8, hidden 128, 6 hidden layers = 3 pre-activation residual
    blocks, Mish, output 2) with fixed random weights - there are no checkpoints offline.  Outputs are clamped to the env's
    action box +-0.01 (pushing.py:203-205), the role the data-derived action bounds play in BC_Agent.predict
    (bc_agent.py:262)."""

    def __init__(self, input_dim=10, hidden_dim=128, num_hidden_layers=6, output_dim=2, seed=0, device="cuda", bound=0.01):
        super().__init__()
        assert num_hidden_layers % 2 == 0
        g = torch.Generator().manual_seed(seed)
        self.bound = bound

        def lin(i, o):
            l = torch.nn.Linear(i, o)
            with torch.no_grad():
                k = 1.0 / i ** 0.5
                l.weight.copy_((torch.rand(o, i, generator=g) * 2 - 1) * k)
                l.bias.copy_((torch.rand(o, generator=g) * 2 - 1) * k)
            return l

        self.inp = lin(input_dim, hidden_dim)
        self.blocks = torch.nn.ModuleList([torch.nn.ModuleList([lin(hidden_dim, hidden_dim), lin(hidden_dim, hidden_dim)])
                                           for _ in range(num_hidden_layers // 2)])
        self.out = lin(hidden_dim, output_dim)
        self.act = torch.nn.Mish()
        self.to(device)

    def reset(self):
        pass

    @torch.no_grad()
    def predict_batch(self, obs: torch.Tensor) -> torch.Tensor:
        x = self.inp(obs.to(torch.float32))
        for l1, l2 in self.blocks:
            x = x + l2(self.act(l1(self.act(x))))
        return self.out(x).clamp_(-self.bound, self.bound)
