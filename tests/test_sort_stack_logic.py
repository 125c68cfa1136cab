"""Sorting / Stacking task logic restated in the oracle ahead of these tasks' physics, pinned against the
reference's own code (tests/golden/ref_sort_stack_task.npz, generator: tests/golden/gen_reference_goldens.py): observations,
success conditions, the order-of-completion mode code with its np.packbits quirk (-1 entries count as set bits) and the
constant pose MjScene returns for boxes that are not in the model; Stacking's colour-order mode string, mean distance and the
three-cubes-stacked success test."""
import os

import numpy as np
import pytest

from oracle.oracle import SortLogic, StackLogic

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_sort_stack_task.npz")


@pytest.mark.parametrize("nb", [2, 4, 6])
def test_sorting_logic(nb):
    g = np.load(G)
    box, rob, obs, succ, code = (g["sort%d_%s" % (nb, k)] for k in ("box", "rob", "obs", "succ", "code"))
    lg = SortLogic(nb)
    for e in range(box.shape[0]):
        lg.reset()
        for t in range(box.shape[1]):
            o, s, c = lg.step(box[e, t], rob[e, t])
            assert s == succ[e, t] and c == code[e, t], (e, t, c, code[e, t])
            assert np.all(np.abs(o - obs[e, t]) <= 2e-6 * np.maximum(1.0, np.abs(obs[e, t])))
    assert (succ.sum() > 0 or nb == 6) and len(np.unique(code)) >= 3


def test_stacking_logic():
    g = np.load(G)
    box, obs, succ, md, modes, target = g["stack_box"], g["stack_obs"], g["stack_succ"], g["stack_md"], g["stack_mode"], g["stack_target"]
    lg = StackLogic()
    for e in range(box.shape[0]):
        lg.reset()
        for t in range(box.shape[1]):
            o, s, d, m = lg.step(box[e, t], target)
            assert s == succ[e, t] and m == str(modes[e, t]), (e, t, m, modes[e, t])
            assert abs(d - md[e, t]) < 1e-15
            assert np.all(np.abs(o - obs[e, t]) <= 2e-6 * np.maximum(1.0, np.abs(obs[e, t])))
    assert succ.sum() > 20 and len(np.unique(modes[:, -1])) >= 4
