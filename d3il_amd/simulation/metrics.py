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
