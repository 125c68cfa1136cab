"""Metric tails of the Sorting and Stacking harnesses from integer count tables, pinned against the
reference's own code (tests/golden/ref_sorting_stacking_metrics.npz, generator: tests/golden/gen_reference_goldens.py)."""
import os

import numpy as np
import pytest

from d3il_amd.simulation.metrics import sorting_metrics, stacking_metrics

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_sorting_stacking_metrics.npz")


def _counts(mode, succ, keys):
    return np.array([[np.sum((mode[c] == k) & (succ[c] == 1)) for k in keys] for c in range(mode.shape[0])])


def test_sorting_metric_tail():
    g = np.load(G)
    me, su, keys = g["sort_mode"], g["sort_succ"], g["sort_keys"]
    sr, ent, kl, score = sorting_metrics(_counts(me, su, keys), int(su.sum()), su.size, me.shape[1], g["sort_prior"])
    assert abs(sr - float(g["sort_success_rate"])) < 1e-7
    assert abs(ent - float(g["sort_entropy"])) < 2e-6 and abs(kl - float(g["sort_KL"])) < 2e-6
    assert abs(score - float(g["sort_score"])) < 2e-6


def test_stacking_metric_tail():
    g = np.load(G)
    nt = g["stack_m3"].shape[1]
    r = stacking_metrics(_counts(g["stack_m1"], g["stack_s1"], range(3)), _counts(g["stack_m2"], g["stack_s2"], range(6)),
                         _counts(g["stack_m3"], g["stack_s3"], range(6)), int(g["stack_s1"].sum()), int(g["stack_s2"].sum()),
                         int(g["stack_s3"].sum()), g["stack_s1"].size, nt, g["stack_prior1"], g["stack_prior2"], g["stack_prior3"])
    for k, v in r.items():
        assert abs(v - float(g["stack_" + k])) < 3e-6, (k, v, float(g["stack_" + k]))
