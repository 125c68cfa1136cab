"""tests/golden/oracle_outcome_sets.json (made by tools/oracle_sensitivity.py, used by tests/test_gpu_count_parity.py): structure, and a live
re-run of the CPU oracle on a few decided contexts - the fixture is an output of oracle/ on committed inputs, nothing else."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fixture():
    with open(os.path.join(ROOT, "tests", "golden", "oracle_outcome_sets.json")) as f:
        return json.load(f)


def test_structure_and_the_plan_of_the_undecided_contexts():
    fx = _fixture()
    assert set(fx) >= {"pushing", "pushing_sampled", "sorting"}
    so = fx["sorting"]["outcomes"]      # Sorting: independent uniform draws in (-1e-9, 1e-9) m on every cube's x, y - the size of the one-step device / oracle difference
    assert sorted(map(int, so)) == list(range(60)) and fx["sorting"]["k"] >= 12 and fx["sorting"]["eps"] <= 1e-9
    assert 1 <= sum(len(o) > 1 for o in so.values()) <= 8
    for task, n in (("pushing", 60), ("pushing_sampled", 120)):
        outs = fx[task]["outcomes"]
        assert sorted(map(int, outs)) == list(range(n)) and fx[task]["k"] >= 24 and fx[task]["eps"] <= 1e-12
        undecided = [int(i) for i, o in outs.items() if len(o) > 1]
        # every context on which the oracle disagrees with itself runs plan 2 of the scripted policy (red cube to the green target first:
        # the cubes cross paths) - VERDICT r3 observed that all device / oracle mismatches were on that plan
        assert undecided and all(i % 4 == 2 for i in undecided)
        assert all(len(o) >= 1 and all(isinstance(s, bool) and isinstance(m, int) for s, m in o) for o in outs.values())


def test_decided_contexts_reproduce_on_the_live_oracle():
    from tests import oracle_episodes as oe
    from tools.oracle_sensitivity import _q0
    from d3il_amd.simulation.pushing_sim import load_test_contexts
    fx = _fixture()["pushing"]["outcomes"]
    ctx, q0 = load_test_contexts(), _q0("pushing")
    for i in (0, 1, 3):
        assert len(fx[str(i)]) == 1
        _, succ, mode, _, _ = oe.pushing_episode((i, ctx[i], q0, 400, i % 4))
        assert [succ, mode] == fx[str(i)][0]
