"""Pushing row of SURVEY 8: task logic + metric tail pinned against the reference's own Python code (goldens made by
tests/golden/gen_reference_goldens.py), and first-principles checks of the free-body / box-contact physics the oracle adds
for this task (parity with MuJoCo itself is unpinned: it is not installed anywhere this runs)."""
import os

import numpy as np
import pytest

from oracle import oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(HERE, "golden", "ref_pushing_task.npz"))


def test_task_logic_matches_reference(push_oracle, gold):
    """obs / success / first-visit mode logic / mean distance / reward of Block_Push_Env (pushing.py:255-280,335-459)."""
    E, T = gold["box"].shape[:2]
    for e in range(E):
        for t in range(T):
            obs, succ, mode, first, md, rew = push_oracle.push_logic(gold["box"][e, t], gold["rob"][e, t], reset=(t == 0))
            assert succ == gold["succ"][e, t] and mode == gold["mode"][e, t] and first == gold["first"][e, t], (e, t)
            ro = gold["obs"][e, t]
            assert np.all(np.abs(obs - ro) <= 2e-6 * np.maximum(1.0, np.abs(ro))), (e, t, obs, ro)
            assert abs(md - gold["mean_distance"][e, t]) < 1e-15 and abs(rew - gold["reward"][e, t]) < 1e-14
    assert set(np.unique(gold["mode"])) == {-1, 0, 1, 2, 3} and gold["succ"].sum() > 100   # the fixture exercises every branch


def test_metric_tail_matches_reference(gold):
    from d3il_amd.simulation.metrics import pushing_metrics
    me, su = gold["metric_mode"], gold["metric_succ"]
    nc, nt = me.shape
    counts = np.array([[np.sum((me[c] == m) & (su[c] == 1)) for m in range(4)] for c in range(nc)])
    sr, ent, _ = pushing_metrics(counts, int(su.sum()), su.size, nt)
    assert abs(sr - float(gold["metric_success_rate"])) < 1e-7
    assert abs(ent - float(gold["metric_entropy"])) < 1e-6


def test_contexts_fixture(push_contexts):
    assert push_contexts.shape == (60, 14)
    assert np.all((push_contexts[:, 0] >= 0.4) & (push_contexts[:, 0] <= 0.5))       # red box space, pushing.py:53-55
    assert np.all((push_contexts[:, 7] >= 0.55) & (push_contexts[:, 7] <= 0.65))     # green box space, pushing.py:56-58
    np.testing.assert_allclose(np.linalg.norm(push_contexts[:, 3:7], axis=1), 1, atol=1e-6)


# ------------------------------------------------------------------------------------------------ collision functions
def _rand_quat(rng):
    q = rng.standard_normal(4)
    return q / np.linalg.norm(q)


def _qmul(a, b):
    return np.array([a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                     a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]])


def _q2m(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def test_box_on_slab_known_answer():
    """A yawed cube sunk 2 mm into a large slab: four contacts at the bottom corners, depth 2 mm, normal = slab normal,
    position midway between the two surfaces."""
    yaw = 0.3
    q = np.array([np.cos(yaw / 2), 0, 0, np.sin(yaw / 2)])
    c = orc.box_box([0.4, 0, -0.02], [1, 0, 0, 0], [0.49, 0.98, 0.001], [0.5, 0.1, 0.009], q, [0.03, 0.03, 0.03])
    assert len(c) == 4
    np.testing.assert_allclose(c[:, 0], -0.002, atol=1e-15)
    np.testing.assert_allclose(c[:, 4:7], [[0, 0, 1]] * 4, atol=1e-15)
    np.testing.assert_allclose(c[:, 3], -0.019 - 0.001, atol=1e-15)
    corners = np.array([[sx * 0.03, sy * 0.03] for sx in (1, -1) for sy in (1, -1)]) @ _q2m(q)[:2, :2].T + [0.5, 0.1]
    got = c[:, 1:3]
    for k in corners:
        assert np.min(np.linalg.norm(got - k, axis=1)) < 1e-14


def test_box_box_is_frame_invariant_and_antisymmetric():
    rng = np.random.default_rng(3)
    n_hit = 0
    for _ in range(300):
        s1, s2 = rng.uniform(0.02, 0.06, 3), rng.uniform(0.02, 0.06, 3)
        q1, q2 = _rand_quat(rng), _rand_quat(rng)
        p1 = rng.uniform(-0.1, 0.1, 3)
        p2 = p1 + rng.uniform(-0.07, 0.07, 3)
        c = orc.box_box(p1, q1, s1, p2, q2, s2)
        if len(c) == 0:
            continue
        n_hit += 1
        assert np.all(c[:, 0] < 0)
        np.testing.assert_allclose(np.linalg.norm(c[:, 4:7], axis=1), 1, atol=1e-12)
        # contact points lie inside both boxes inflated by the penetration
        for box_p, box_q, box_s in ((p1, q1, s1), (p2, q2, s2)):
            loc = (c[:, 1:4] - box_p) @ _q2m(box_q)
            assert np.all(np.abs(loc) <= box_s + np.abs(c[:, :1]) + 1e-12)
        # rigid motion of the pair moves the contacts with it
        qg, tg = _rand_quat(rng), rng.uniform(-1, 1, 3)
        Rg = _q2m(qg)
        c2 = orc.box_box(Rg @ p1 + tg, _qmul(qg, q1), s1, Rg @ p2 + tg, _qmul(qg, q2), s2)
        assert len(c2) == len(c)
        np.testing.assert_allclose(c2[:, 0], c[:, 0], atol=1e-12)
        np.testing.assert_allclose(c2[:, 1:4], c[:, 1:4] @ Rg.T + tg, atol=1e-11)
        np.testing.assert_allclose(c2[:, 4:7], c[:, 4:7] @ Rg.T, atol=1e-11)
        # swapping the arguments flips the normal and keeps depth (same separating axis is found)
        c3 = orc.box_box(p2, q2, s2, p1, q1, s1)
        assert len(c3) >= 1
        np.testing.assert_allclose(c3[0, 4:7], -c[0, 4:7], atol=1e-9)
        np.testing.assert_allclose(np.min(c3[:, 0]), np.min(c[:, 0]), atol=1e-9)
    assert n_hit > 100


def test_box_box_separated_boxes_do_not_collide():
    rng = np.random.default_rng(4)
    for _ in range(200):
        s1, s2 = rng.uniform(0.02, 0.06, 3), rng.uniform(0.02, 0.06, 3)
        d = rng.standard_normal(3)
        d *= (np.linalg.norm(s1) + np.linalg.norm(s2) + 1e-3) / np.linalg.norm(d)   # farther than the circumscribed spheres
        assert len(orc.box_box([0, 0, 0], _rand_quat(rng), s1, d, _rand_quat(rng), s2)) == 0


def test_cylinder_box_matches_brute_force():
    """(axis segment)-box distance against dense sampling of the segment."""
    rng = np.random.default_rng(5)
    n_hit = 0
    for _ in range(300):
        sb = rng.uniform(0.02, 0.05, 3)
        qb, qc = _rand_quat(rng), _rand_quat(rng)
        pb = rng.uniform(-0.1, 0.1, 3)
        pc = pb + rng.uniform(-0.12, 0.12, 3)
        rad, half = 0.01, 0.15
        Rb, Rc = _q2m(qb), _q2m(qc)
        t = np.linspace(-half, half, 20001)
        pts = pc + t[:, None] * Rc[:, 2]
        loc = (pts - pb) @ Rb
        dd = np.linalg.norm(loc - np.clip(loc, -sb, sb), axis=1)
        dmin = dd.min()
        r = orc.cyl_box(pc, qc, rad, half, pb, qb, sb, margin=0.05)
        if dmin < 1e-6:           # axis inside the box: only sign/finite checks
            assert r is not None and r[0] < -rad + 1e-9
            continue
        if dmin - rad >= 0.05:
            assert r is None
            continue
        n_hit += 1
        assert r is not None
        assert abs(r[0] - (dmin - rad)) < 2e-6        # sampling resolution
        np.testing.assert_allclose(np.linalg.norm(r[4:7]), 1, atol=1e-12)
        # the contact position sits half-way between the two surfaces along the normal
        locp = (r[1:4] - pb) @ Rb
        surf = locp - 0.5 * r[0] * (r[4:7] @ Rb)
        assert np.max(np.abs(surf) - sb) < 1e-9
    assert n_hit > 100


def test_cylinder_parallel_to_face_uses_middle_of_overlap():
    # upright rod next to an upright box: the overlap of the rod with the box's height is [0.002, 0.03] -> contact at its middle
    r = orc.cyl_box([0.045, 0.0, 0.152], [1, 0, 0, 0], 0.01, 0.15, [0, 0, 0], [1, 0, 0, 0], [0.03, 0.03, 0.03], margin=0.01)
    np.testing.assert_allclose(r[0], 0.005, atol=1e-15)
    np.testing.assert_allclose(r[4:7], [1, 0, 0], atol=1e-15)
    np.testing.assert_allclose(r[1:4], [0.0325, 0.0, 0.5 * (0.002 + 0.03)], atol=1e-12)


# ------------------------------------------------------------------------------------------------ free-body physics
def _settle(o, ctx, init_qpos, n=400):
    qpos = np.zeros(23)
    qpos[0:14] = ctx
    qpos[14:21] = init_qpos
    o.set_state(qpos, np.zeros(21))
    for _ in range(n):
        o.mj_step()
    return o.state()


def test_box_rest_height_closed_form(push_oracle, pushing_json, init_qpos, push_contexts):
    """SURVEY 8c KAT (iv): a box resting on the table sinks in until four corner contacts carry m g.
    With reference acceleration -k d r (solref -> k, solimp -> d) and regulariser R = (1 - d)/(d m) per normal row, the force of
    one contact at rest is k d^2 |r| m / (1 - d); 4 f = m g gives |r| = g (1 - d) / (4 k d^2)."""
    ctx = push_contexts[0].copy()
    ctx[2] = ctx[9] = 0.011      # start close to rest
    qpos, qvel = _settle(push_oracle, ctx, init_qpos)
    assert np.max(np.abs(qvel[:12])) < 1e-9
    # mixed contact parameters (equal solmix): mean of box (0.02 1 | 0.9 0.95 0.001 0.5 2) and table (0.002 1 | 0.999 0.999 0.001)
    tc, dr = 0.5 * (0.02 + 0.002), 1.0
    d0, dw, width = 0.5 * (0.9 + 0.999), 0.5 * (0.95 + 0.999), 0.001
    k = 1.0 / (dw * dw * tc * tc * dr * dr)
    r = 1e-5
    for _ in range(50):
        x = r / width
        y = 2.0 * x * x if x < 0.5 else 1 - 2.0 * (1 - x) ** 2     # power 2, midpoint 0.5
        d = d0 + y * (dw - d0)
        r = 9.81 * (1 - d) / (4 * k * d * d)
    table_top = -0.02 + 0.001
    for a in (0, 7):
        np.testing.assert_allclose(qpos[a + 2] - 0.03, table_top - r, atol=1e-9)
    assert len(push_oracle.contacts()) == 8


def test_free_boxes_conserve_momentum_in_collision(push_oracle, init_qpos):
    """Two cubes collide in free fall-less space: contact forces are internal, so linear and angular momentum are conserved
    (cube inertia is isotropic, so L = m x cross v + I R omega_body)."""
    o = push_oracle
    o.set_gravity([0, 0, 0])
    qpos = np.zeros(23)
    qpos[14:21] = init_qpos
    qpos[0:7] = [0.45, 0.0, 0.5, np.cos(0.2), 0, np.sin(0.2), 0]
    qpos[7:14] = [0.53, 0.01, 0.52, np.cos(0.4), 0, 0, np.sin(0.4)]
    qvel = np.zeros(21)
    qvel[0:6] = [0.5, 0, 0, 0.3, -0.2, 0.1]
    qvel[6:12] = [-0.5, 0.05, 0, 0, 0.4, 0]
    o.set_state(qpos, qvel)
    m, I = 0.05, 3e-5

    def momenta():
        qp, qv = o.state()
        P, L = np.zeros(3), np.zeros(3)
        for a, d in ((0, 0), (7, 6)):
            x, v = qp[a:a + 3], qv[d:d + 3]
            R = _q2m(qp[a + 3:a + 7])
            P += m * v
            L += m * np.cross(x, v) + I * (R @ qv[d + 3:d + 6])
        return P, L

    P0, L0 = momenta()
    hit = 0
    for _ in range(150):
        o.mj_step()
        hit += len(o.contacts()) > 0
    P1, L1 = momenta()
    assert hit > 3
    qp, qv = o.state()
    assert abs(qv[0] - 0.5) > 0.05                 # the collision did change the velocities
    np.testing.assert_allclose(P1, P0, atol=1e-12)
    np.testing.assert_allclose(L1, L0, atol=1e-11)


def test_pushing_rollout_moves_the_box(push_oracle, init_qpos, push_contexts):
    """End-to-end sanity of Block_Push_Env.reset/step in the oracle: the rod reaches the red box and pushes it in +y; the
    box stays on the table; obs/done/info follow the protocol."""
    o = push_oracle
    o.env_start(init_qpos)
    obs = o.push_reset(push_contexts[0])
    assert obs.shape == (8,) and obs.dtype == np.float32
    np.testing.assert_allclose(obs[2:4], push_contexts[0][0:2], atol=1e-6)
    box_y0 = float(obs[3])
    des = obs[:2].astype(float).copy()
    z = 0.12235931
    for t in range(60):
        d = obs[2:4].astype(float) - des
        n = np.linalg.norm(d)
        des = des + d / max(n, 1e-9) * min(0.006, n)
        obs, rew, done, info = o.push_step(np.concatenate([des, [z], [0, 1, 0, 0]]))
        assert not done and info["mode"] == -1 and not info["success"]
        assert rew < 0
    qp, _ = o.state()
    assert float(obs[3]) > box_y0 + 0.1
    assert 0.009 < qp[2] < 0.014 and 0.0109 < qp[9] < 0.0111    # pushed box may tilt slightly, the other one rests
