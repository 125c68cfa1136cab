"""Generic engine of the Sorting kernel (d3il_amd/csrc/gen_step.h) built for the host, against the oracle: scene constants,
the reset transient, and a scripted push that takes a cube over the platform edge into its bin (same trajectory, same
observation, same completion-order mode code)."""
import numpy as np
import pytest

from oracle.oracle import Oracle
from tests.hostcheck.hostcheck import GenHostCheck
from tests.test_sorting_oracle import CTX, sort_blob, sort_init_qpos  # noqa: F401  (fixtures)


def _state_err(h, o):
    qp, qv = o.state()
    e = 0.0
    for b in range(h.nb):
        p, q, v = h.box(b)
        e = max(e, np.abs(p - qp[7 * b:7 * b + 3]).max(), np.abs(q - qp[7 * b + 3:7 * b + 7]).max(), 1e-2 * np.abs(v - qv[6 * b:6 * b + 6]).max())
    return max(e, np.abs(h.s[:9] - qp[7 * h.nb:7 * h.nb + 9]).max())


def test_scene_constants(sort_blob):
    h = GenHostCheck(sort_blob)
    assert (h.nb, h.ns, h.ns_core) == (4, 19, 11)
    st = h.statics
    # table_plane and support_body precede the cubes in the model (geom 1 of a pair), walls and platform follow them
    assert st[:11, 6].tolist() == [1, 1] + [0] * 9
    # the frame around the table top (lab_surrounding.xml:3-114): four upper beams and four posts, all before the cubes in the model, their
    # tops 1 mm BELOW the table top (z = -0.02 against -0.019) and reaching 2 cm beyond its edge: what a cube tipping over the edge lands
    # on; in the list after everything that stands inside the table
    rim = st[11:]
    assert rim[:, 6].tolist() == [1] * 8
    np.testing.assert_allclose(rim[:, 2] + rim[:, 5], -0.02, atol=1e-12)
    assert sorted(np.round(rim[:, 3:5].max(axis=1), 3).tolist()) == [0.02] * 4 + [0.47] * 2 + [0.96] * 2
    np.testing.assert_allclose(h.inner, [-0.01, -0.9, 0.81, 0.9], atol=1e-12)      # table footprint shrunk by 8 cm
    np.testing.assert_allclose(h.outer, [-0.11, -1.0, 0.91, 1.0], atol=1e-12)      # outer faces of the frame
    np.testing.assert_allclose(st[0, :6], [0.4, 0, -0.02, 0.49, 0.98, 0.001], atol=1e-12)
    np.testing.assert_allclose(st[10, :6], [0.5, -0.1, 0.0, 0.3, 0.3, 0.1], atol=1e-12)
    assert sorted(np.round(st[2:10, 3:5].min(axis=1), 3).tolist()) == [0.005] * 6 + [0.01] * 2


def test_reset_and_hop(sort_blob, sort_init_qpos):
    o = Oracle(sort_blob)
    o.env_start(sort_init_qpos)
    h = GenHostCheck(sort_blob)
    obs_o = o.sort_reset(CTX)
    obs_h = h.reset(sort_init_qpos, CTX)
    assert obs_h.shape == (14,) and np.array_equal(obs_o, obs_h)
    z = float(o.body(sort_blob.tcp_body)[0][2])
    a = np.concatenate([obs_o[:2].astype(float), [z], [0, 1, 0, 0]])
    for t in range(14):                               # cubes leave the platform box through its top, fly ~5 cm, land
        oo, do, io = o.sort_step(a)
        oh, dh, ih = h.step(a, fast=bool(t % 2))
        assert _state_err(h, o) < 1e-12 and np.array_equal(oo, oh) and do == dh and io["mode"] == ih["mode"] == 240
        assert not (ih["flags"] & 0x1F0000)           # no solver failure, overflow or off-table flag


def test_scripted_push_into_the_bin(sort_blob, sort_init_qpos):
    o = Oracle(sort_blob)
    o.env_start(sort_init_qpos)
    h = GenHostCheck(sort_blob)
    obs = o.sort_reset(CTX)
    h.reset(sort_init_qpos, CTX)
    z = float(o.body(sort_blob.tcp_body)[0][2])
    des = obs[:2].astype(float)
    code = 240
    for t in range(170):
        box = obs[2:4].astype(float)
        if t < 12:
            target = des.copy()
        else:
            aligned = abs(des[0] - box[0]) < 0.008 and des[1] < box[1] - 0.02
            target = np.array([box[0], 0.36]) if aligned else box + np.array([0.0, -0.06])
        d = target - des
        n = np.linalg.norm(d)
        des = des + d / max(n, 1e-9) * min(0.006, n)
        a = np.concatenate([des, [z], [0, 1, 0, 0]])
        obs, done, info = o.sort_step(a)
        oh, dh, ih = h.step(a)
        # rod on cube, cube sliding over the platform edge and dropping between the bin walls: two f64 formulations of the
        # same solve drift apart slowly (cf. Pushing); the bound is generous against the observed 2e-8
        assert _state_err(h, o) < 1e-6 and np.abs(obs - oh).max() < 1e-6 and done == dh and info["mode"] == ih["mode"]
        assert not (ih["flags"] & 0x1F0000)
        code = info["mode"]
        if code != 240:
            break
    assert code == 0b01110000 and ih["success"] is False


def test_sorting_2_scene_tracks_the_oracle(sort_init_qpos):
    """The two-box variant of the scene (sorting.py:121-138) on the same engine: reset, landing and the first pushes."""
    from d3il_amd.model import blob
    m2 = blob.load("sorting_2")
    assert m2.n_obj == 2
    o = Oracle(m2)
    o.env_start(sort_init_qpos)
    h = GenHostCheck(m2)
    assert (h.nb, h.ns, h.ns_core, h.n) == (2, 19, 11, 42 + 26 + 21 + 2)
    ctx = CTX[[0, 2]]
    obs_o, obs_h = o.sort_reset(ctx), h.reset(sort_init_qpos, ctx)
    assert obs_h.shape == (8,) and np.array_equal(obs_o, obs_h)
    z = float(o.body(m2.tcp_body)[0][2])
    des = obs_o[:2].astype(float)
    for t in range(60):
        box = obs_o[2:4].astype(float)
        if t >= 12:
            aligned = abs(des[0] - box[0]) < 0.008 and des[1] < box[1] - 0.02
            target = np.array([box[0], 0.36]) if aligned else box + np.array([0.0, -0.06])
            d = target - des
            n = np.linalg.norm(d)
            des = des + d / max(n, 1e-9) * min(0.006, n)
        a = np.concatenate([des, [z], [0, 1, 0, 0]])
        obs_o, do, io = o.sort_step(a)
        oh, dh, ih = h.step(a)
        assert _state_err(h, o) < 1e-7 and np.abs(obs_o - oh).max() < 1e-6 and do == dh
        assert io["mode"] == ih["mode"] == 0b11000000 and not (ih["flags"] & 0x1F0000)      # two unset entries


def test_cube_over_the_table_edge_meets_the_frame_beam(sort_blob, sort_init_qpos):
    """VERDICT r3 missing #3: cube <-> frame-beam pairs (lab_surrounding.xml:3-114).  A cube dropped with its centre 1 cm beyond the table's
    front edge comes down on the table edge, the support block and the front beam (top 1 mm below the table top, reaching 2 cm beyond its
    edge), is kicked outwards by the push-out of its deep first penetration and slides off the frame.  The host build of the kernel math
    evaluates the beam pair like the oracle - same trajectory while the beam carries the cube - and raises OFF_TABLE exactly when the
    cube centre has passed the outer face of the frame (x > 0.91), not, as before, 8 cm inside the table edge."""
    o = Oracle(sort_blob)
    o.env_start(sort_init_qpos)
    h = GenHostCheck(sort_blob)
    ctx = CTX.copy()
    ctx[0, :2] = [0.90, 0.35]                         # table edge at x = 0.89, beam 0.87 .. 0.91
    ctx[1, :2] = [0.86, -0.55]                        # 3 cm inside the edge (inside the old 8 cm margin): table only, stays
    obs_o, obs_h = o.sort_reset(ctx), h.reset(sort_init_qpos, ctx)
    assert np.array_equal(obs_o, obs_h)
    z = float(o.body(sort_blob.tcp_body)[0][2])
    a = np.concatenate([obs_o[:2].astype(float), [z], [0, 1, 0, 0]])
    beam_steps = 0
    for t in range(12):
        oo, do, io = o.sort_step(a)
        oh, dh, ih = h.step(a)
        con = o.contacts()
        beam_steps += any(int(row[8]) == 2 for row in con)          # geom 2 = front_upper against a cube
        assert _state_err(h, o) < 1e-7, (t, _state_err(h, o))
        x0 = h.box(0)[0][0]
        off = bool(ih["flags"] & (1 << 19))
        assert (off or x0 <= h.outer[2]) and (not off or x0 > h.outer[2] - 0.01), (t, x0, off)      # raised at the outer face of the frame, not before
        assert not (ih["flags"] & ((1 << 16) | (1 << 18)))
    assert beam_steps >= 3, "the cube should have been carried by the front beam for a few steps"
    assert h.box(0)[0][0] > 0.95 and abs(h.box(1)[0][0] - 0.86) < 1e-3 and abs(h.box(1)[0][2] - 0.011) < 1e-3


def test_rod_on_two_cubes(sort_blob, sort_init_qpos):
    """A context in which the scripted push brings the rod against two cubes at once (env steps 73 ..): the island {arm, two cubes} is not one the tree
    solver takes (two rod contacts: the six rows of two points on one rigid rod are dependent) - the joint solver runs next to the tree solver.
    Same trajectory as the oracle's at the conditioning level of the contact regime."""
    torch = pytest.importorskip("torch")
    from d3il_amd.agents import ScriptedPushPolicy
    from d3il_amd.envs.sorting import sample_contexts
    from tests.hostcheck.hostcheck import lib, _p
    ctx = sample_contexts(60, 4, seed=0)[1]
    o = Oracle(sort_blob)
    o.env_start(sort_init_qpos)
    h = GenHostCheck(sort_blob)
    obs = o.sort_reset(ctx.reshape(4, 7))
    h.reset(sort_init_qpos, ctx)
    pol = ScriptedPushPolicy("sorting", device="cpu")
    des, z = np.array(obs[:2], dtype=float), float(h.s[27])
    hist = np.zeros(80, dtype=np.int64)
    lib().hc_gen_island_hist(_p(hist), 1)
    worst = 0.0
    for t in range(82):
        des = des + pol.predict_batch(torch.as_tensor(np.concatenate([des, obs.astype(float)])[None]))[0].numpy()
        a = np.concatenate([des, [z], [0, 1, 0, 0]])
        obs, do, io = o.sort_step(a)
        oh, dh, ih = h.step(a)
        worst = max(worst, _state_err(h, o))
        assert not (ih["flags"] & 0x1F0000) and np.allclose(obs, oh, atol=1e-6, rtol=0) and io["mode"] == ih["mode"]
    lib().hc_gen_island_hist(_p(hist), 0)
    assert worst < 2e-8, worst
    assert int(hist[:36].sum()) > 0, hist[:36]           # the stretch does exercise the joint solver
