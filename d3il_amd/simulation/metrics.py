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


def mode_entropy_kl(mode_counts, n_trajectories_per_context: int, prior, n_mode: int | None = None):
    """Behaviour entropy and KL divergence to a prior over modes, from integer counts.

    The formula shared by ``Sorting_Sim.test_agent`` (simulation/sorting_sim.py:194-208) and ``Stacking_Sim.cal_KL``
    (simulation/stacking_sim.py:143-167): p(m | c) = count / n_trajectories, row-normalised (+1e-12), contexts without a
    single counted rollout dropped, entropy = -sum p log(p + 1e-12) / log(n_mode) averaged over the remaining contexts,
    KL = -entropy - mean_c sum_m p log(prior + 1e-12) / log(n_mode).  float32 torch arithmetic like the reference.
    mode_counts[c][m]: successful rollouts of context c whose mode is the m-th key.
    """
    import torch

    counts = torch.as_tensor(np.asarray(mode_counts), dtype=torch.int64)
    n_mode = counts.shape[1] if n_mode is None else n_mode
    mode_probs = (counts / n_trajectories_per_context).to(torch.float32)
    mode_probs = mode_probs / (mode_probs.sum(1).reshape(-1, 1) + 1e-12)
    mode_probs = mode_probs[torch.nonzero(mode_probs.sum(1), as_tuple=True)[0]]
    prior = torch.as_tensor(np.asarray(prior))
    logn = torch.log(torch.tensor(n_mode))
    entropy = -(mode_probs * torch.log(mode_probs + 1e-12) / logn).sum(1).mean()
    log_ = (mode_probs * torch.log(prior + 1e-12) / logn).sum(1).mean()
    return float(entropy), float(-entropy - log_)


def sorting_metrics(mode_counts, n_success: int, n_rollouts: int, n_trajectories_per_context: int, prior):
    """Metric tail of ``Sorting_Sim.test_agent`` (sorting_sim.py:191-213): success rate, entropy, KL, score = success - KL.
    mode_counts[c][k] counts successful rollouts whose ``info['mode']`` (np.packbits code, sorting.py:461-463) equals the
    k-th key of the mode-prior table."""
    import torch

    success_rate = float(torch.tensor(float(n_success), dtype=torch.float32) / n_rollouts)
    entropy, kl = mode_entropy_kl(mode_counts, n_trajectories_per_context, prior)
    return success_rate, entropy, kl, success_rate - kl


def stacking_metrics(counts_1, counts_2, counts_3, n_success_1: int, n_success_2: int, n_success_3: int, n_rollouts: int,
                     n_trajectories_per_context: int, prior_1, prior_2, prior_3):
    """Metric tail of ``Stacking_Sim.test_agent`` (stacking_sim.py:226-248): success rates for >= 1, >= 2 and 3 stacked boxes,
    entropy / KL of the 3-, 6- and 6-way mode tables against the priors built at stacking_sim.py:47-62; score = sum of the
    three success rates."""
    import torch

    def rate(k):
        return float(torch.tensor(float(k), dtype=torch.float32) / n_rollouts)

    r1, r2, r3 = rate(n_success_1), rate(n_success_2), rate(n_success_3)
    e1, k1 = mode_entropy_kl(counts_1, n_trajectories_per_context, prior_1, 3)
    e2, k2 = mode_entropy_kl(counts_2, n_trajectories_per_context, prior_2, 6)
    e3, k3 = mode_entropy_kl(counts_3, n_trajectories_per_context, prior_3, 6)
    return dict(successes=r3, successes_1_box=r1, successes_2_boxes=r2, entropy_1=e1, KL_1=k1, entropy_2=e2, KL_2=k2,
                entropy_3=e3, KL_3=k3, score=r1 + r2 + r3)


def aligning_metrics(mode_counts, n_success: int, n_rollouts: int, n_trajectories_per_context: int, distance_sum: float = 0.0):
    """Metric tail of ``Aligning_Sim.test_agent`` (simulation/aligning_sim.py:160-204) from integer counts: the Pushing formulas with TWO behaviour
    modes (0: the rod pushed from inside the box walls, 1: from outside; aligning.py:288-312).  Returns what the reference logs:
    (score = 0.5 (success rate + entropy), success rate, entropy, mean distance, mode_probs)."""
    success_rate, entropy, mode_probs = pushing_metrics(mode_counts, n_success, n_rollouts, n_trajectories_per_context, n_modes=2)
    return 0.5 * (success_rate + entropy), success_rate, entropy, float(distance_sum) / max(n_rollouts, 1), mode_probs
