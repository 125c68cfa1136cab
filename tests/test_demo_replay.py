"""Replay of recorded reference demonstrations through the oracle: the hook that can pin the physics rows.

Skipped unless ``tests/golden/demo_avoiding.npz`` exists (made by tools/make_demo_fixtures.py from the reference's demonstration
pickles, which are not part of its source tree).  The recorded command trace des_c_pos[t] is fed to the oracle env the way the
teleoperation / rollout loop feeds env.step (avoiding_sim.py:61-66: action = [des_x, des_y, fixed_z, 0, 1, 0, 0]); the TCP trace the
oracle produces must follow the MuJoCo-recorded c_pos within the north star's 1e-4.  The log may have been written before or after
the physics of a step, so both alignments are tried and the better one is asserted."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = os.path.join(ROOT, "tests", "golden", "demo_avoiding.npz")


@pytest.mark.skipif(not os.path.exists(FIX), reason="no demonstration fixture (tools/make_demo_fixtures.py needs the reference's demo data)")
def test_oracle_replays_recorded_avoiding_demonstrations(avoiding_blob, init_qpos):
    from oracle.oracle import Oracle
    g = np.load(FIX)
    worst = 0.0
    for i in range(int(g["n_demos"])):
        des, cpos = g["demo%d_des_c_pos" % i], g["demo%d_c_pos" % i]
        o = Oracle(avoiding_blob)
        o.env_start(init_qpos)
        o.env_reset()
        s, _ = o.env_state()
        z = s[27]
        trace = [s[25:28].copy()]
        for t in range(len(des)):
            o.env_step(np.array([des[t, 0], des[t, 1], z, 0, 1, 0, 0]))
            s, _ = o.env_state()
            trace.append(s[25:28].copy())
        trace = np.array(trace)
        errs = []
        for lag in (0, 1):
            m = min(len(cpos), len(trace) - lag)
            errs.append(np.abs(trace[lag:lag + m, :2] - cpos[:m, :2]).max())
        worst = max(worst, min(errs))
    assert worst < 1e-4, worst
