"""Host build of the Pushing kernel math (d3il_amd/csrc/push_step.h via tests/hostcheck) against the oracle.

The two are independent formulations of the same sub-step: the oracle is a generic world-frame engine (dense constraint
Jacobian, dense Cholesky, cold-started Newton), the kernel math is specialised (link-frame arm dynamics, sparse contact
rows, skyline Cholesky of the [cube|cube|arm] Hessian, warm-started Newton).  Agreement to ~1e-9 over thousands of
sub-steps therefore checks both.  No GPU needed; the GPU run of the same code is covered by tests/test_gpu_parity_pushing.py.
"""
import numpy as np
import pytest

from oracle.oracle import Oracle
from tests.hostcheck.hostcheck import PushHostCheck


@pytest.fixture(scope="module")
def push_hc(pushing_blob):
    return PushHostCheck(pushing_blob)


def _chase(obs, des, step=0.006):
    """scripted policy: move the desired TCP towards the red cube (then through it)"""
    d = obs[2:4].astype(float) - des
    n = np.linalg.norm(d)
    return des + d / max(n, 1e-9) * min(step, n)


def test_reset_matches_oracle_on_all_reference_contexts(push_oracle, push_hc, init_qpos, push_contexts):
    push_oracle.env_start(init_qpos)
    for ctx in push_contexts:
        obs_o = push_oracle.push_reset(ctx)
        obs_h = push_hc.reset(init_qpos, ctx)
        np.testing.assert_array_equal(obs_o, obs_h)
        so, fo = push_oracle.push_state()
        np.testing.assert_allclose(push_hc.s[:68], so, atol=1e-11, rtol=0)
        assert fo[6] >= 16          # the cubes start sunk into both table slabs (8 contacts each)


@pytest.mark.parametrize("ctx_id", [0, 7, 23])
def test_pushing_rollout_matches_oracle(push_oracle, push_hc, init_qpos, push_contexts, ctx_id):
    ctx = push_contexts[ctx_id]
    push_oracle.env_start(init_qpos)
    obs_o = push_oracle.push_reset(ctx)
    obs_h = push_hc.reset(init_qpos, ctx)
    des = obs_o[:2].astype(float)
    z = float(push_hc.s[27])        # TCP z after reset (D3IL_STATE_TCP + 2)
    moved = False
    for t in range(45):
        des = _chase(obs_o, des)
        a = np.concatenate([des, [z], [0, 1, 0, 0]])
        obs_o, rew_o, done_o, info_o = push_oracle.push_step(a)
        obs_h, rew_h, done_h, info_h = push_hc.step(a)
        so, fo = push_oracle.push_state()
        sh = push_hc.s[:68]
        assert done_o == done_h and info_o["mode"] == info_h["mode"] and info_o["success"] == info_h["success"]
        assert not (info_h["flags"] & ((1 << 16) | (1 << 18) | (1 << 19))), hex(info_h["flags"])   # solver fail / overflow / off table
        np.testing.assert_allclose(obs_h, obs_o, atol=1e-6, rtol=1e-6)
        # positions (arm q, cube pos/quat, TCP) to 1e-7, velocities to 1e-4 (cube angular velocity is the sensitive quantity
        # while the cube rocks on its contacts)
        pos_idx = list(range(0, 9)) + list(range(25, 28)) + list(range(42, 49)) + list(range(55, 62))
        vel_idx = list(range(9, 18)) + list(range(49, 55)) + list(range(62, 68))
        np.testing.assert_allclose(sh[pos_idx], so[pos_idx], atol=1e-7, rtol=0)
        np.testing.assert_allclose(sh[vel_idx], so[vel_idx], atol=1e-4, rtol=0)
        assert abs(rew_o - rew_h) < 1e-7 and abs(info_o["mean_distance"] - info_h["mean_distance"]) < 1e-7
        moved = moved or abs(so[43] - ctx[1]) > 0.02
    assert moved        # the rod did push the red cube


def test_slow_ik_path_gives_the_same_rollout(push_hc, pushing_blob, init_qpos, push_contexts):
    other = PushHostCheck(pushing_blob)
    ctx = push_contexts[3]
    o1, o2 = push_hc.reset(init_qpos, ctx), other.reset(init_qpos, ctx)
    des = o1[:2].astype(float)
    z = float(push_hc.s[27])
    for t in range(10):
        des = _chase(o1, des)
        a = np.concatenate([des, [z], [0, 1, 0, 0]])
        o1 = push_hc.step(a, fast=True)[0]
        o2 = other.step(a, fast=False)[0]
    np.testing.assert_allclose(push_hc.s[:68], other.s[:68], atol=1e-8)


def _special_states(s0):
    """mid-episode states that force the rarely taken solver paths"""
    tcp = s0[25:28]
    out = {}
    s = s0.copy(); s[3] = -0.06                                       # joint 4 beyond its upper limit (-0.0698): memory-resident path
    out["arm_limit"] = s
    s = s0.copy()                                                     # rod squeezed between the cubes: memory-resident path
    s[42:45] = [tcp[0] - 0.0395, tcp[1], 0.011]; s[45:49] = [1, 0, 0, 0]; s[49:55] = 0
    s[55:58] = [tcp[0] + 0.0395, tcp[1], 0.011]; s[58:62] = [1, 0, 0, 0]; s[62:68] = 0
    out["rod_on_both"] = s
    s = s0.copy()                                                     # rod on cube 1, cube 1 pressed against cube 2: coupled path, all couplings
    s[42:45] = [tcp[0] + 0.0395, tcp[1], 0.011]; s[45:49] = [1, 0, 0, 0]; s[49:55] = 0
    s[55:58] = [tcp[0] + 0.0395 + 0.0598, tcp[1] + 0.004, 0.011]; s[58:62] = [np.cos(0.1), 0, 0, np.sin(0.1)]; s[62:68] = 0
    out["rod_and_cube_cube"] = s
    return out


def test_rare_solver_paths_match_oracle(push_oracle, push_hc, init_qpos, push_contexts):
    o, hc = push_oracle, push_hc
    o.env_start(init_qpos)
    obs = o.push_reset(push_contexts[0])
    hc.reset(init_qpos, push_contexts[0])
    a = np.concatenate([obs[:2].astype(float), [0.12235931], [0, 1, 0, 0]])
    for t in range(12):
        o.push_step(a)
    s0, _ = o.push_state()
    for name, s in _special_states(s0).items():
        o.push_set_state(s, step=12, terminated=False, first_visit=-1, ik_valid=True)
        hc.s[:68] = s; hc.s[68:] = 0; hc.f[0] = 1 << 15; hc.f[1] = 12
        o.push_step(a)
        so, fo = o.push_state()
        _, _, _, ih = hc.step(a)
        assert not (ih["flags"] & ((1 << 16) | (1 << 18))), name
        assert fo[6] >= 4, name
        np.testing.assert_allclose(hc.s[:68], so, atol=1e-6, rtol=0, err_msg=name)


def test_pushing_the_green_cube_matches_oracle(push_oracle, push_hc, init_qpos, push_contexts):
    """The rod on cube 1 exercises the cube-group permutation of the coupled solver (the cube under the rod becomes group 0)."""
    o, hc = push_oracle, push_hc
    ctx = push_contexts[5]
    o.env_start(init_qpos)
    obs = o.push_reset(ctx)
    hc.reset(init_qpos, ctx)
    des = obs[:2].astype(float)
    moved = False
    pos_idx = list(range(0, 9)) + list(range(25, 28)) + list(range(42, 49)) + list(range(55, 62))
    for t in range(45):
        d = obs[5:7].astype(float) - des
        n = np.linalg.norm(d)
        des = des + d / max(n, 1e-9) * min(0.006, n)
        a = np.concatenate([des, [0.12235931], [0, 1, 0, 0]])
        obs, _, _, io = o.push_step(a)
        _, _, _, ih = hc.step(a)
        so, fo = o.push_state()
        assert not (ih["flags"] & ((1 << 16) | (1 << 18) | (1 << 19)))
        np.testing.assert_allclose(hc.s[:68][pos_idx], so[pos_idx], atol=1e-7, rtol=0)
        moved = moved or abs(so[56] - ctx[8]) > 0.02
    assert moved


def test_line_search_regression_state(push_oracle, push_hc, init_qpos):
    """A mid-push state (recorded from a GPU rollout) where phi'(alpha) of the coupled solve is sigmoid-like: Newton on alpha
    alone jumps between the two flat sides for ever; the bracket-halving safeguard must converge (no solver-fail flag)."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "push_hard_state.npz"))
    s, fl, step, a = g["state"], int(g["flags"]), int(g["step"]), g["action"]
    push_hc.s[:68] = s; push_hc.s[68:] = 0
    push_hc.f[0] = fl & ~(1 << 6)                # no warm start: the recorded one is not part of the fixture
    push_hc.f[1] = step
    _, _, _, info = push_hc.step(a)
    assert not (info["flags"] & (1 << 16))
    push_oracle.env_start(init_qpos)
    push_oracle.push_set_state(s, step=step, terminated=False, first_visit=(fl & 7) - 1, ik_valid=True)
    push_oracle.push_step(a)
    so, fo = push_oracle.push_state()
    np.testing.assert_allclose(push_hc.s[:68], so, atol=1e-7, rtol=0)


def test_device_collision_routines_equal_the_oracle_ones():
    """box_box / cyl_box of push_step.h (register-only variants: sorting network, unrolled scans) against the oracle's."""
    import ctypes as C
    from oracle import oracle as orc
    from tests.hostcheck.hostcheck import lib, _p
    L = lib()
    rng = np.random.default_rng(9)

    def rq():
        q = rng.standard_normal(4)
        return q / np.linalg.norm(q)

    n_c = n_b = 0
    for it in range(600):
        sb = rng.uniform(0.02, 0.05, 3)
        qb, qc = rq(), rq()
        if it % 3 == 0:                     # upright rod next to an upright yawed cube: the regime of the task (parallel axes)
            yaw = rng.uniform(-np.pi, np.pi)
            qb, qc = np.array([np.cos(yaw / 2), 0, 0, np.sin(yaw / 2)]), np.array([0.0, 1, 0, 0])
        pb = rng.uniform(-0.1, 0.1, 3)
        pc = pb + rng.uniform(-0.08, 0.08, 3) + np.array([0, 0, 0.12]) * (it % 3 == 0)
        a = [np.ascontiguousarray(v, float) for v in (pc, qc, pb, qb, sb)]
        out = np.zeros(7)
        hit = L.hc_cyl_box(_p(a[0]), _p(a[1]), C.c_double(0.01), C.c_double(0.15), _p(a[2]), _p(a[3]), _p(a[4]), C.c_double(0.02), _p(out))
        ref = orc.cyl_box(pc, qc, 0.01, 0.15, pb, qb, sb, margin=0.02)
        assert bool(hit) == (ref is not None)
        if hit:
            n_c += 1
            np.testing.assert_allclose(out, ref, atol=1e-12)
        s1, s2 = rng.uniform(0.02, 0.06, 3), rng.uniform(0.02, 0.06, 3)
        q1, q2 = rq(), rq()
        p1 = rng.uniform(-0.1, 0.1, 3)
        p2 = p1 + rng.uniform(-0.07, 0.07, 3)
        b = [np.ascontiguousarray(v, float) for v in (p1, q1, s1, p2, q2, s2)]
        out2 = np.zeros((16, 7))
        n = L.hc_box_box(*[_p(v) for v in b], C.c_double(0.0), _p(out2))
        ref2 = orc.box_box(p1, q1, s1, p2, q2, s2)
        assert n == len(ref2)
        if n:
            n_b += 1
            np.testing.assert_allclose(out2[:n], ref2, atol=1e-12)
    assert n_c > 100 and n_b > 100


def test_box_box_face_clipping_regimes_equal_the_oracle():
    """The register-only box_box (select-chain Sutherland-Hodgman, no private arrays) against the oracle's array version in the regimes the
    random poses above seldom reach: a box lying on a box with a small offset / yaw / tilt (incident face partly outside the reference face:
    the clip path, 4 - 8 contacts), equal faces exactly on each other (the finger tips), boxes a hair apart inside the margin."""
    import ctypes as C
    from oracle import oracle as orc
    from tests.hostcheck.hostcheck import lib, _p
    L = lib()
    rng = np.random.default_rng(21)

    def quat(axis, ang):
        axis = np.asarray(axis, float) / np.linalg.norm(axis)
        return np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * axis])

    def qmul(a, b):
        return np.array([a[0] * b[0] - a[1:] @ b[1:], *(a[0] * b[1:] + b[0] * a[1:] + np.cross(a[1:], b[1:]))])

    counts = np.zeros(9, dtype=int)
    for it in range(1500):
        s1 = rng.uniform(0.02, 0.06, 3) if it % 4 else np.array([0.03, 0.03, 0.03])
        s2 = rng.uniform(0.02, 0.06, 3) if it % 4 else np.array([0.03, 0.05, 0.03])
        if it % 5 == 0:
            s2 = s1.copy()
        margin = [0.0, 0.001, 0.002][it % 3]
        q1 = quat([0, 0, 1], rng.uniform(-np.pi, np.pi))
        tilt = quat(rng.standard_normal(3), rng.uniform(0, [0.0, 1e-3, 0.05][it % 3]))
        q2 = qmul(tilt, quat([0, 0, 1], rng.uniform(-np.pi, np.pi) if it % 7 else 0.0))
        if it % 7 == 0:
            q2 = q1.copy()
        p1 = rng.uniform(-0.1, 0.1, 3)
        gap = rng.uniform(-0.004, 0.0015)
        p2 = p1 + np.array([rng.uniform(-0.05, 0.05), rng.uniform(-0.05, 0.05), s1[2] + s2[2] + gap])
        if it % 11 == 0:
            p2[:2] = p1[:2]
        b = [np.ascontiguousarray(v, float) for v in (p1, q1, s1, p2, q2, s2)]
        out = np.zeros((16, 7))
        n = L.hc_box_box(*[_p(v) for v in b], C.c_double(margin), _p(out))
        ref = orc.box_box(p1, q1, s1, p2, q2, s2, margin=margin)
        assert n == len(ref), (it, n, len(ref))
        counts[n] += 1
        if n:
            np.testing.assert_allclose(out[:n], ref, atol=1e-12)
    assert counts[4] > 100 and counts[5:].sum() > 100 and counts[1] > 10, counts


@pytest.mark.parametrize("ctx_id", [0, 7, 23])
def test_pushing_on_the_generic_engine_matches_oracle(push_oracle, pushing_blob, init_qpos, push_contexts, ctx_id):
    """The Pushing task on the generic engine (gen_step.h: GEN_TASK_PUSHING - the product's default engine for the task since round 5: its two cubes, the
    table slabs and the frame beams as static boxes, the tree solver) on the host: same state layout as the Pushing engine (rows 0 .. 88), reward and
    info['mean_distance'] in the two task rows behind it, first-visit / mode bits in the flag word; the same rollout as the oracle's."""
    from tests.hostcheck.hostcheck import GenHostCheck
    h = GenHostCheck(pushing_blob)
    assert (h.nb, h.n_obs) == (2, 8) and h.n == 42 + 26 + 21 + 2 and h.ns > h.ns_core == 2      # the two slabs inside the table, the frame beams behind them
    ctx = push_contexts[ctx_id]
    push_oracle.env_start(init_qpos)
    obs_o = push_oracle.push_reset(ctx)
    obs_h = h.reset(init_qpos, ctx)
    np.testing.assert_array_equal(obs_o, obs_h)
    so, fo = push_oracle.push_state()
    np.testing.assert_allclose(h.s[:68], so, atol=1e-11, rtol=0)
    des = obs_o[:2].astype(float)
    z = float(h.s[27])
    moved = False
    for t in range(45):
        des = _chase(obs_o, des)
        a = np.concatenate([des, [z], [0, 1, 0, 0]])
        obs_o, rew_o, done_o, info_o = push_oracle.push_step(a)
        obs_h, done_h, info_h = h.step(a)
        so, fo = push_oracle.push_state()
        sh = h.s[:68]
        mode_h = info_h["mode"] if info_h["mode"] < 32768 else info_h["mode"] - 65536
        assert done_o == done_h and info_o["mode"] == mode_h and info_o["success"] == info_h["success"]
        assert not (info_h["flags"] & ((1 << 16) | (1 << 18) | (1 << 19))), hex(info_h["flags"])
        np.testing.assert_allclose(obs_h, obs_o, atol=1e-6, rtol=1e-6)
        pos_idx = list(range(0, 9)) + list(range(25, 28)) + list(range(42, 49)) + list(range(55, 62))
        vel_idx = list(range(9, 18)) + list(range(49, 55)) + list(range(62, 68))
        np.testing.assert_allclose(sh[pos_idx], so[pos_idx], atol=1e-7, rtol=0)
        np.testing.assert_allclose(sh[vel_idx], so[vel_idx], atol=1e-4, rtol=0)
        assert abs(rew_o - h.s[90]) < 1e-7 and abs(info_o["mean_distance"] - h.s[89]) < 1e-7      # the task rows: info['mean_distance'], reward
        moved = moved or abs(so[43] - ctx[1]) > 0.02
    assert moved
