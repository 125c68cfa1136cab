"""Stacking on the CPU: the oracle's Stacking env (physical sanity of the grasp contacts) and the host build of the device engine
(d3il_amd/csrc/stack_step.h via tests/hostcheck) against it.

The two are independent formulations of the same sub-step: the oracle is the generic world-frame engine (dense 27-dof constraint
Jacobian, dense Cholesky, libccd-style MPR on generic geoms), the device engine is specialised (link-frame arm dynamics, contact
rows rebuilt from records, block-sparse Cholesky).  A scripted pick-and-place exercises finger-tip (box-box) and finger-hull (MPR)
grasp contacts with condim 4, box-on-table and box-on-box contacts.  No GPU needed; the GPU run of the same header is covered by
tests/test_gpu_parity_stacking.py.
"""
import numpy as np
import pytest

from d3il_amd.controllers.offline_ik import offline_ik
from d3il_amd.controllers.scripted_stacking import build_trajectory
from d3il_amd.kinematics import UrdfChain
from d3il_amd.model import blob as blob_mod
from oracle.oracle import Oracle
from tests.hostcheck.hostcheck import StackHostCheck


@pytest.fixture(scope="module")
def stack_js():
    return blob_mod.load_json("stacking")


@pytest.fixture(scope="module")
def stack_blob(stack_js):
    return blob_mod.pack(stack_js)


@pytest.fixture(scope="module")
def stack_init_qpos(stack_js):
    c, tc = stack_js["controller"], stack_js["task_const"]
    q, it, err = offline_ik(UrdfChain(stack_js["urdf_chain"]), c["default_qpos"], list(tc["init_end_eff_pos"]) + list(tc["init_end_eff_quat"]),
                            np.array(c["joint_pos_min"]), np.array(c["joint_pos_max"]))
    # the reference's own offline IK on this start pose: 72 iterations
    assert it == 72 and err < 1e-5
    np.testing.assert_allclose(q, [0, -0.043619, 0, -2.188421, 0, 2.149904, 0.785397], atol=2e-6)
    return q


@pytest.fixture(scope="module")
def stack_contexts():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "d3il_amd", "data", "stacking_test_contexts.npy"))


def test_model_pairs_and_consts(stack_js, stack_blob):
    o = Oracle(stack_blob)
    assert (o.nq, o.nv) == (30, 27)
    names = [g["name"] for g in stack_js["geoms"]]
    pairs = {tuple(sorted((names[a], names[b]))) for a, b in o.pairs(True)}
    # finger <-> box, finger <-> finger and box <-> box pairs are evaluated
    for f in ("panda_rb0_leftfinger:geom2", "finger1_rb0_tip_collision", "panda_rb0_rightfinger:geom2", "finger2_rb0_tip_collision"):
        for bx in ("red_box:geom", "green_box:geom", "blue_box:geom"):
            assert tuple(sorted((f, bx))) in pairs
    assert ("finger1_rb0_tip_collision", "finger2_rb0_tip_collision") in pairs and ("panda_rb0_leftfinger:geom2", "panda_rb0_rightfinger:geom2") in pairs
    hc = StackHostCheck(stack_blob)      # build_stack_consts accepts the model
    assert hc.n == 94          # 67 state rows + the solver's warm start


def test_boxes_settle_at_their_rest_height(stack_blob, stack_init_qpos, stack_contexts):
    """A box dropped 11 mm into the table (contexts write z = 0, stacking.py:99-125) is pushed out and rests where its weight balances
    the soft contact: the same closed form as for the Pushing cubes (all three boxes weigh 50 g and lie on a 6 cm-high face)."""
    o = Oracle(stack_blob)
    o.env_start(stack_init_qpos)
    o.stack_reset(stack_contexts[3])
    for _ in range(8):
        obs, done, info = o.stack_step(np.concatenate([stack_init_qpos, [1.0]]))
    z = obs[[2, 6, 10]]
    assert np.all(np.abs(z - 0.010984) < 2e-5), z
    assert abs(o.stack_robot_state()[7] - 0.08) < 5e-4          # open_fingers: both fingers at their 0.04 m stops (still settling)


@pytest.mark.parametrize("ctx_id", [0, 3])
def test_host_engine_tracks_oracle_through_pick_and_place(stack_js, stack_blob, stack_init_qpos, stack_contexts, ctx_id):
    o = Oracle(stack_blob)
    o.env_start(stack_init_qpos)
    hc = StackHostCheck(stack_blob)
    ctx = stack_contexts[ctx_id]
    obs_o, obs_h = o.stack_reset(ctx), hc.reset(stack_init_qpos, ctx)
    np.testing.assert_array_equal(obs_o, obs_h)
    np.testing.assert_allclose(hc.s[:67], o.stack_state(), atol=1e-11, rtol=0)
    traj = build_trajectory(stack_js, stack_init_qpos, ctx, n_boxes=1, speed=0.8)
    worst, held_z, n_grasp = 0.0, 0.0, 0
    for a in traj:
        obs_o, done_o, info_o = o.stack_step(a)
        obs_h, done_h, info_h = hc.step(a)
        assert done_o == done_h and info_o["mode"] == info_h["mode"] and info_o["success"] == info_h["success"]
        assert not (info_h["flags"] & ((1 << 16) | (1 << 18) | (1 << 19)))      # solver failure / contact overflow / off table
        worst = max(worst, np.abs(hc.s[:67] - o.stack_state()).max())
        xyz = [0, 1, 2, 4, 5, 6, 8, 9, 10]
        np.testing.assert_allclose(obs_h[xyz], obs_o[xyz], atol=2e-6, rtol=1e-5)
        np.testing.assert_allclose(np.arctan(obs_h[3::4]), np.arctan(obs_o[3::4]), atol=1e-5)     # tan(yaw) is unbounded near 90 degrees
        cons = o.contacts()
        if sum(1 for c in cons if c[9] > 40) >= 4:      # finger geoms (ids > 40) on the box: tip boxes (up to 4 each) and hulls (1 each)
            n_grasp += 1
            held_z = max(held_z, float(obs_o[2]))
    # positions and velocities of arm and boxes over ~120 env steps x 30 sub-steps incl. the grasp, the lift and the release
    assert worst < 2e-6, worst
    assert n_grasp > 20 and held_z > 0.08                                   # the red box was lifted in a multi-contact grasp
    assert info_o["mode"] == "r" and np.hypot(obs_o[0] - 0.5, obs_o[1] - 0.2) < 0.06   # and put down in the target zone (pos_min_dist)


def test_empty_gripper_closes_on_itself(stack_blob, stack_init_qpos, stack_contexts):
    """close_fingers with nothing between the fingers: the finger-finger pairs (tip boxes, finger hulls) carry the closing force;
    host engine and oracle agree and the gripper width ends at the contact equilibrium, not at the joint stops."""
    o = Oracle(stack_blob)
    o.env_start(stack_init_qpos)
    hc = StackHostCheck(stack_blob)
    o.stack_reset(stack_contexts[1]); hc.reset(stack_init_qpos, stack_contexts[1])
    a = np.concatenate([stack_init_qpos, [0.0]])
    worst = 0.0
    for t in range(14):
        o.stack_step(a); hc.step(a)
        worst = max(worst, np.abs(hc.s[:67] - o.stack_state()).max())
    assert worst < 1e-7, worst
    w = o.stack_robot_state()[7]
    assert 0.0 < w < 0.002, w
    kinds = {(int(c[8]), int(c[9])) for c in o.contacts()}
    assert len([k for k in kinds if k[0] > 40 and k[1] > 40]) >= 1           # at least one robot-robot (finger-finger) pair in contact


def test_palm_on_a_box_hand_hull_contact(stack_js, stack_blob, stack_init_qpos, stack_contexts):
    """The box <-> hand pair (panda_invisible.xml:72: panda_hand:geom2, convex hull of handv.stl, 773 vertices; VERDICT r2 missing #1): the open
    gripper is lowered over the red box until the palm presses on its top.  The oracle (generic MPR over the hull) and the host build of
    the device engine (bounding-box cull + MPR) agree through approach, contact (about 1 mm deep), hold and retreat."""
    from d3il_amd.controllers.scripted_stacking import build_palm_press
    names = [g["name"] for g in stack_js["geoms"]]
    hand_g = names.index("panda_rb0_hand:geom2")
    o = Oracle(stack_blob)
    assert {tuple(sorted((names[a], names[b]))) for a, b in o.pairs(True)} >= {("panda_rb0_hand:geom2", "red_box:geom"), ("green_box:geom", "panda_rb0_hand:geom2")}
    o.env_start(stack_init_qpos)
    hc = StackHostCheck(stack_blob)
    ctx = stack_contexts[0]
    o.stack_reset(ctx); hc.reset(stack_init_qpos, ctx)
    worst, n_hand, deepest = 0.0, 0, 0.0
    for a in build_palm_press(stack_js, stack_init_qpos, ctx):
        o.stack_step(a)
        _, _, ih = hc.step(a)
        assert not (ih["flags"] & ((1 << 16) | (1 << 18) | (1 << 19) | (1 << 20)))
        worst = max(worst, np.abs(hc.s[:67] - o.stack_state()).max())
        hand = [c for c in o.contacts() if int(c[8]) == hand_g or int(c[9]) == hand_g]
        if hand:
            n_hand += 1
            deepest = min(deepest, min(c[0] for c in hand))
    assert n_hand >= 15 and deepest < -5e-4, (n_hand, deepest)
    assert worst < 1e-6, worst


def test_host_engine_reads_nothing_it_has_not_written(stack_blob, stack_init_qpos, stack_contexts):
    """Read-before-write detector of the one-lane engine (hc_stack_poison: both scratch areas hold NaN before every env step; the warm start is carried in
    the state rows): an empty gripper closing on itself - the finger <-> finger pairs - and the first steps of a grasp give the same states as without
    the poison, bit for bit, and stay finite."""
    from tests.hostcheck.hostcheck import lib
    a = np.concatenate([stack_init_qpos, [0.0]])
    runs = []
    for poison in (0, 1):
        lib().hc_stack_poison(poison)
        try:
            hc = StackHostCheck(stack_blob)
            hc.reset(stack_init_qpos, stack_contexts[1])
            states = []
            for t in range(12):
                hc.step(a)
                states.append(hc.s.copy())
            runs.append(np.stack(states))
        finally:
            lib().hc_stack_poison(0)
    assert np.isfinite(runs[1]).all() and np.array_equal(runs[0], runs[1])

