"""Pushing: the HIP path through the C ABI against the CPU oracle.

Contact-rich pushing is chaotic: two correct f64 implementations starting 1e-12 apart separate by a factor of ~1.5-2 per
env step while the cube rocks on its contacts.  Parity is therefore asserted (a) at reset, (b) over a bounded horizon, and
(c) as ONE-STEP parity from identical mid-episode states sampled along GPU rollouts (the oracle is loaded with the GPU
state, both take the same action, the results must agree) - that covers every regime without compounding chaos.
Integer outputs (done, success, mode, first-visit logic) are compared exactly.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
POS = list(range(0, 9)) + list(range(25, 28)) + list(range(42, 49)) + list(range(55, 62))
VEL = list(range(9, 18)) + list(range(49, 55)) + list(range(62, 68))
BAD = (1 << 16) | (1 << 18) | (1 << 19)          # solver fail, contact overflow, off table


@pytest.fixture(scope="module")
def ctx60():
    return np.load(os.path.join(ROOT, "d3il_amd", "data", "pushing_test_contexts.npy"))


def _env(n, **kw):
    from d3il_amd.envs.pushing import BlockPushVecEnv
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    return BlockPushVecEnv(n, device=0, **kw)


def _chase(env, des, step=0.006):
    o64 = env.obs.to(torch.float64)
    d = o64[:, 2:4] - des
    nn = d.norm(dim=1, keepdim=True).clamp_min(1e-9)
    return des + d / nn * torch.minimum(nn, torch.full_like(nn, step))


def _action(des, z):
    n = des.shape[0]
    quat = torch.tensor([0.0, 1, 0, 0], dtype=torch.float64, device=des.device).expand(n, 4)
    return torch.cat([des, z, quat], dim=1).contiguous()


def test_data_contexts_match_the_reference_fixture(ctx60, push_contexts):
    np.testing.assert_array_equal(ctx60, push_contexts)


def test_reset_matches_oracle_on_all_reference_contexts(ctx60, init_qpos, pushing_blob):
    from oracle.oracle import Oracle
    env = _env(60)
    env.set_init_qpos(init_qpos)
    obs = env.reset(context=ctx60).cpu().numpy()
    st, fl, sc = env.get_state()
    o = Oracle(pushing_blob)
    o.env_start(init_qpos)
    for e in range(60):
        oo = o.push_reset(ctx60[e])
        so, fo = o.push_state()
        np.testing.assert_array_equal(obs[e], oo)
        np.testing.assert_allclose(st[:68, e], so, atol=1e-10, rtol=0)
        assert sc[e] == 0 and not (fl[e] & BAD)
    env.close()


@pytest.mark.parametrize("fast", [1, 0])
def test_bounded_horizon_rollout_matches_oracle(ctx60, init_qpos, pushing_blob, fast):
    from oracle.oracle import Oracle
    n = 96
    env = _env(n)
    env.set_option("ik_fast_path", fast)
    env.set_init_qpos(init_qpos)
    ctx = ctx60[np.arange(n) % 60]
    env.reset(context=ctx)
    check = [0, 23, 24, 47, 59, 95]
    oracles = []
    for e in check:
        o = Oracle(pushing_blob); o.env_start(init_qpos); o.push_reset(ctx[e]); oracles.append(o)
    des = env.robot_state()[:, :2].clone()
    z = env.robot_state()[:, 2:3].clone()
    for t in range(34 if fast else 12):
        des = _chase(env, des)
        act = _action(des, z)
        obs, rew, done, info = env.step(act)
        torch.cuda.synchronize()
        st, fl, sc = env.get_state()
        a = act.cpu().numpy()
        for k, e in enumerate(check):
            oo, ro, do, io = oracles[k].push_step(a[e])
            so, fo = oracles[k].push_state()
            assert not (fl[e] & BAD)
            np.testing.assert_allclose(st[POS, e], so[POS], atol=1e-6, rtol=0, err_msg="t %d env %d" % (t, e))
            np.testing.assert_allclose(obs[e].cpu().numpy(), oo, atol=2e-6, rtol=1e-5)
            assert bool(done[e]) == do and int(info["mode"][e]) == io["mode"] and bool(info["success"][e]) == io["success"]
            assert abs(float(rew[e]) - ro) < 1e-6 and abs(float(info["mean_distance"][e]) - io["mean_distance"]) < 1e-6
    env.close()


def test_north_star_horizon_of_free_running_rollouts(ctx60, init_qpos, pushing_blob):
    """Free-running rollouts (no state re-synchronisation) against the oracle, ALL state rows incl. velocities: the north star's
    1e-4 must hold through reset transient, approach and the first pushes (>= 40 env steps = 1400 sub-steps in every followed
    environment); the horizon at which each threshold is first exceeded is recorded.  Beyond it the two are different valid
    rollouts of a chaotic contact system (a rocking cube amplifies 1e-16 by x1.5-2 per env step)."""
    from oracle.oracle import Oracle
    n = 120
    env = _env(n)
    env.set_init_qpos(init_qpos)
    ctx = ctx60[np.arange(n) % 60]
    env.reset(context=ctx)
    follow = {}
    for e in (0, 7, 33, 61, 90, 119):
        o = Oracle(pushing_blob); o.env_start(init_qpos); o.push_reset(ctx[e]); follow[e] = o
    horizon = {e: {} for e in follow}
    des = env.robot_state()[:, :2].clone()
    z = env.robot_state()[:, 2:3].clone()
    T = 80
    for t in range(T):
        des = _chase(env, des)
        act = _action(des, z)
        env.step(act)
        torch.cuda.synchronize()
        st, fl, sc = env.get_state()
        assert not (fl & BAD).any()
        a = act.cpu().numpy()
        for e, o in follow.items():
            o.push_step(a[e])
            so, _ = o.push_state()
            err = float(np.abs(st[:68, e] - so).max())
            for thr in (1e-8, 1e-6, 1e-4):
                if err > thr and thr not in horizon[e]:
                    horizon[e][thr] = t
    h4 = [horizon[e].get(1e-4, T) for e in follow]
    print("pushing horizons (1e-8 / 1e-6 / 1e-4): %s" % {e: tuple(horizon[e].get(k, T) for k in (1e-8, 1e-6, 1e-4)) for e in follow})
    assert min(h4) >= 40, horizon
    env.close()


@pytest.mark.parametrize("strict", [0, 1])
def test_one_step_parity_from_mid_episode_states(ctx60, init_qpos, pushing_blob, strict):
    """States sampled along a GPU rollout (rest, rod contact, cube-cube contact) are loaded into the oracle; one env step
    (35 sub-steps) with the same action must agree on ALL state rows, velocities included, far inside the north star's 1e-4.

    What bounds the agreement: not the device solvers' stopping rule - the run with the oracle's rule
    (solver_strict = 1) gives the same numbers - but the conditioning of the soft-contact problem itself: contact rows carry
    D ~ 1e6 .. 1e7 against a cube inertia of 3e-5 kg m^2, so a round-off level residual (1e-8 N, the size of the oracle's OWN
    optimality residual at its solution, tests/test_parity_conditioning.py) moves the cube's angular acceleration by 1e-6 .. 1e-5
    rad/s^2: 3.5e-8 rad/s per sub-step, 2 .. 4e-7 per env step between any two correct f64 implementations."""
    from oracle.oracle import Oracle
    n = 120
    env = _env(n)
    env.set_option("solver_strict", strict)
    env.set_init_qpos(init_qpos)
    ctx = ctx60[np.arange(n) % 60]
    env.reset(context=ctx)
    o = Oracle(pushing_blob)
    o.env_start(init_qpos)
    des = env.robot_state()[:, :2].clone()
    z = env.robot_state()[:, 2:3].clone()
    rng = np.random.default_rng(0)
    n_contact = n_bb = checked = 0
    for t in range(70):
        if t < 45:
            des = _chase(env, des)                      # go to the red cube and push it
        else:                                           # then push on towards the green cube
            o64 = env.obs.to(torch.float64)
            d = o64[:, 5:7] - des
            nn = d.norm(dim=1, keepdim=True).clamp_min(1e-9)
            des = des + d / nn * torch.minimum(nn, torch.full_like(nn, 0.006))
        act = _action(des, z)
        torch.cuda.synchronize()
        st0, fl0, sc0 = env.get_state()
        obs, rew, done, info = env.step(act)
        torch.cuda.synchronize()
        st1, fl1, sc1 = env.get_state()
        if t < 20 or t % 2:
            continue
        a = act.cpu().numpy()
        for e in rng.choice(n, 6, replace=False):
            first = int(fl0[e] & 7) - 1
            o.push_set_state(st0[:68, e], step=sc0[e], terminated=bool(fl0[e] & (1 << 12)), first_visit=first, ik_valid=bool(fl0[e] & (1 << 15)))
            oo, ro, do, io = o.push_step(a[e])
            so, fo = o.push_state()
            ncon = len(o.contacts())
            geoms = o.contacts()[:, 8:10] if ncon else np.zeros((0, 2))
            assert not (fl1[e] & BAD), hex(fl1[e])
            np.testing.assert_allclose(st1[POS, e], so[POS], atol=2e-8, rtol=0, err_msg="t %d env %d" % (t, e))
            np.testing.assert_allclose(st1[VEL, e], so[VEL], atol=2e-6, rtol=0, err_msg="t %d env %d" % (t, e))
            assert bool(done[e]) == do and int(info["mode"][e]) == io["mode"] and bool(info["success"][e]) == io["success"]
            assert int(fl1[e] & 7) - 1 == fo[3]
            checked += 1
            n_contact += ncon > 8 or (ncon > 0 and ncon != 8)
    assert checked > 100 and n_contact > 20
    env.close()


def test_full_size_batch_properties(ctx60, init_qpos):
    """4096 environments = 69 tiles of the 60 contexts: identical contexts + identical actions must give bit-identical
    results in every wave / lane; physical invariants hold; the episode cap sets done."""
    n = 4096
    env = _env(n, max_steps_per_episode=12)
    env.set_init_qpos(init_qpos)
    ctx = ctx60[np.arange(n) % 60]
    env.reset(context=ctx)
    des = env.robot_state()[:, :2].clone()
    z = env.robot_state()[:, 2:3].clone()
    for t in range(12):
        des = _chase(env, des, step=0.02)
        obs, rew, done, info = env.step(_action(des, z))
        torch.cuda.synchronize()
        assert int(done.sum()) == (n if t == 11 else 0)
    st, fl, sc = env.get_state()
    assert not np.any(fl & BAD)
    ref = st[:68, :60]
    for tile in range(1, n // 60):
        assert np.array_equal(st[:68, 60 * tile:60 * tile + 60], ref), tile
    pos, quat = env.box_state()
    pos, quat = pos.cpu().numpy(), quat.cpu().numpy()
    np.testing.assert_allclose(np.linalg.norm(quat, axis=2), 1, atol=1e-12)
    assert np.all((pos[:, :, 2] > 0.005) & (pos[:, :, 2] < 0.02))          # cubes rest on the table top (z_rest = 0.01098)
    assert np.all(sc == 12)
    env.close()


def test_ragged_sizes_masks_and_errors(ctx60, init_qpos):
    from d3il_amd import capi
    for n in (1, 25, 100):
        env = _env(n)
        with pytest.raises(capi.D3ilError):
            env.reset(context=ctx60[np.arange(n) % 60])          # env.start() first
        env.set_init_qpos(init_qpos)
        with pytest.raises(ValueError):
            env.reset()                                          # Block_Push_Env.reset(random=False) needs a context
        ctx = ctx60[np.arange(n) % 60]
        env.reset(context=ctx)
        des = env.robot_state()[:, :2].clone()
        z = env.robot_state()[:, 2:3].clone()
        for t in range(3):
            des = _chase(env, des)
            env.step(_action(des, z))
        torch.cuda.synchronize()
        st, fl, sc = env.get_state()
        assert np.all(sc == 3)
        # masked reset: only env 0 goes back to its context
        mask = torch.zeros(n, dtype=torch.uint8, device=env.device); mask[0] = 1
        env.reset(mask=mask, context=ctx)
        torch.cuda.synchronize()
        st2, fl2, sc2 = env.get_state()
        assert sc2[0] == 0 and np.all(sc2[1:] == 3)
        np.testing.assert_array_equal(st2[:68, 1:], st[:68, 1:])
        np.testing.assert_allclose(st2[42:44, 0], ctx[0, 0:2], atol=1e-3)
        env.close()


def test_rare_solver_paths_one_step(ctx60, init_qpos, pushing_blob):
    """Arm joint beyond its limit / rod squeezed between both cubes (memory-resident solver) and rod + cube-cube contact
    (coupled solver with every coupling): states written with set_state, one step against the oracle."""
    from oracle.oracle import Oracle
    from tests.test_push_kernel_host import _special_states
    o = Oracle(pushing_blob)
    o.env_start(init_qpos)
    obs = o.push_reset(ctx60[0])
    a = np.concatenate([obs[:2].astype(float), [0.12235931], [0, 1, 0, 0]])
    for t in range(12):
        o.push_step(a)
    s0, _ = o.push_state()
    cases = _special_states(s0)
    names = list(cases)
    n = 48
    env = _env(n)
    env.set_init_qpos(init_qpos)
    env.reset(context=ctx60[np.zeros(n, dtype=int)])
    st, fl, sc = env.get_state()
    for e in range(n):
        st[:68, e] = cases[names[e % 3]]
        st[68:, e] = 0
    fl[:] = 1 << 15
    sc[:] = 12
    env.set_state(st, fl, sc)
    act = torch.as_tensor(np.tile(a, (n, 1)), dtype=torch.float64, device=env.device).contiguous()
    env.step(act)
    torch.cuda.synchronize()
    st1, fl1, sc1 = env.get_state()
    for k, name in enumerate(names):
        o.push_set_state(cases[name], step=12, terminated=False, first_visit=-1, ik_valid=True)
        o.push_step(a)
        so, fo = o.push_state()
        for e in range(k, n, 3):
            assert not (fl1[e] & BAD), (name, hex(fl1[e]))
            np.testing.assert_allclose(st1[:68, e], so, atol=1e-6, rtol=0, err_msg=name)
    env.close()


def test_success_and_first_visit_mode_logic_on_device(ctx60, init_qpos, pushing_blob):
    """Cubes are placed into the target zones with set_state: first-visit bookkeeping, mode, success, done and mean distance
    of the device path against the oracle for all four visiting orders."""
    from oracle.oracle import Oracle
    o = Oracle(pushing_blob)
    o.env_start(init_qpos)
    obs = o.push_reset(ctx60[0])
    a = np.concatenate([obs[:2].astype(float), [0.12235931], [0, 1, 0, 0]])
    for t in range(8):
        o.push_step(a)
    s0, _ = o.push_state()
    t1, t2 = np.array([0.42, 0.3]), np.array([0.63, 0.3])
    # (first cube, its zone, second cube, its zone): rr->gg, gg->rr, rg->gr, gr->rg  => modes 0, 1, 2, 3
    orders = [((0, t1), (1, t2)), ((1, t2), (0, t1)), ((0, t2), (1, t1)), ((1, t1), (0, t2))]
    n = len(orders)
    env = _env(n)
    env.set_init_qpos(init_qpos)
    env.reset(context=ctx60[np.zeros(n, dtype=int)])
    st, fl, sc = env.get_state()
    oracles = []
    for e, ((c1, z1), _) in enumerate(orders):
        s = s0.copy()
        s[42 + 13 * c1:44 + 13 * c1] = z1 + [0.01, -0.015]; s[44 + 13 * c1] = 0.011
        st[:68, e] = s; st[68:, e] = 0
        oo = Oracle(pushing_blob); oo.env_start(init_qpos)
        oo.push_set_state(s, step=8, terminated=False, first_visit=-1, ik_valid=True)
        oracles.append(oo)
    fl[:] = 1 << 15; sc[:] = 8
    env.set_state(st, fl, sc)
    act = torch.as_tensor(np.tile(a, (n, 1)), dtype=torch.float64, device=env.device).contiguous()
    expected_first = [0, 3, 1, 2]
    for phase in range(3):
        obs, rew, done, info = env.step(act)
        torch.cuda.synchronize()
        st, fl, sc = env.get_state()
        for e in range(n):
            oo, ro, do, io = oracles[e].push_step(a)
            so, fo = oracles[e].push_state()
            assert bool(done[e]) == do and int(info["mode"][e]) == io["mode"] and bool(info["success"][e]) == io["success"], (phase, e)
            assert int(fl[e] & 7) - 1 == fo[3] == expected_first[e]
            assert abs(float(info["mean_distance"][e]) - io["mean_distance"]) < 1e-6
        if phase == 0:       # now bring the second cube into its zone, in both implementations
            for e, (_, (c2, z2)) in enumerate(orders):
                st[42 + 13 * c2:44 + 13 * c2, e] = z2 + [-0.012, 0.01]; st[44 + 13 * c2, e] = 0.011
                st[49 + 13 * c2:55 + 13 * c2, e] = 0
                first = int(fl[e] & 7) - 1
                oracles[e].push_set_state(st[:68, e], step=sc[e], terminated=False, first_visit=first, ik_valid=True)
            env.set_state(st, fl, sc)
        if phase == 1:
            # both cubes are in their zones when the step starts, so is_finished (evaluated before the physics) already reports done
            assert [int(m) for m in info["mode"].cpu()] == [0, 1, 2, 3] and bool(info["success"].all()) and bool(done.all())
    env.close()


def test_random_contexts_and_large_batch(init_qpos):
    """reset(random=True) draws contexts from the spaces of BlockContextManager.sample; a 16384-env batch steps cleanly."""
    from d3il_amd.envs.pushing import sample_contexts
    c = sample_contexts(500, seed=3)
    assert np.all((c[:, 0] >= 0.4) & (c[:, 0] <= 0.5) & (c[:, 7] >= 0.55) & (c[:, 7] <= 0.65))
    assert np.all((c[:, [1, 8]] >= -0.15) & (c[:, [1, 8]] <= 0.0)) and np.all(c[:, [2, 9]] == 0)
    np.testing.assert_allclose(np.linalg.norm(c[:, 3:7], axis=1), 1, atol=1e-12)
    n = 16384
    env = _env(n)
    env.set_init_qpos(init_qpos)
    obs = env.reset(random=True)
    torch.cuda.synchronize()
    assert obs.shape == (n, 8) and bool(torch.isfinite(obs).all())
    des = env.robot_state()[:, :2].clone()
    z = env.robot_state()[:, 2:3].clone()
    for t in range(3):
        des = _chase(env, des)
        obs, rew, done, info = env.step(_action(des, z))
    torch.cuda.synchronize()
    st, fl, sc = env.get_state()
    assert np.isfinite(st).all() and not np.any(fl & BAD) and np.all(sc == 3)
    env.close()
