"""Metric formulas of the rollout harness, computed from integer counts (bit-exact, order independent).

Avoiding (simulation/avoiding_sim.py:126-135): success rate = mean(successes); behaviour entropy over the
distinct 9-bit mode codes of the *successful* rollouts, log base 24.
"""
from __future__ import annotations

import numpy as np


def avoiding_metrics(n_rollouts: int, n_success: int, hist512: np.ndarray):
    """hist512[c] = number of successful rollouts whose mode code (sum_i bit_i << i) is c."""
    success_rate = float(n_success) / float(n_rollouts)
    counts = np.asarray(hist512, dtype=np.int64)
    counts = counts[counts > 0]
    if counts.sum() == 0:
        return success_rate, 0.0
    mode_dist = counts / np.sum(counts)
    entropy = -np.sum(mode_dist * (np.log(mode_dist) / np.log(24)))
    return success_rate, float(entropy)


def pushing_metrics(mode_counts, n_success: int, n_rollouts: int, n_trajectories_per_context: int, n_modes: int = 4):
    """Metric tail of ``Pushing_Sim.test_agent`` (simulation/pushing_sim.py:140-167) from integer counts.

    mode_counts[c][m] = number of *successful* rollouts of context c whose final ``info['mode']`` is m
    (m in 0..3; rollouts ending with mode -1 count towards the success rate only).  The reference evaluates
    these formulas in float32 torch; the same dtype is used here.
    """
    import torch

    counts = torch.as_tensor(np.asarray(mode_counts), dtype=torch.int64)
    mode_probs = (counts / n_trajectories_per_context).to(torch.float32)
    mode_probs = mode_probs / (mode_probs.sum(1).reshape(-1, 1) + 1e-12)
    entropy = -(mode_probs * torch.log(mode_probs + 1e-12) / torch.log(torch.tensor(n_modes))).sum(1).mean()
    success_rate = float(torch.tensor(float(n_success), dtype=torch.float32) / n_rollouts)
    return success_rate, float(entropy), mode_probs
