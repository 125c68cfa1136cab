"""Sorting-4: the HIP path through the C ABI against the CPU oracle.

Like Pushing, contact-rich sorting separates two f64 implementations over long horizons, so parity is asserted at reset,
through the reset transient (cubes leave the platform box through its top and land on it), along scripted pushes that take
a cube over the platform edge into a bin (bounded tolerance), and as ONE-STEP parity from mid-episode device states.
Integer outputs (done, success, completion-order mode code) are compared exactly.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BAD = (1 << 16) | (1 << 18) | (1 << 19)          # solver fail, contact overflow, off table
NB = 4


@pytest.fixture(scope="module")
def sort_blob():
    from d3il_amd.model import blob
    return blob.load("sorting")


@pytest.fixture(scope="module")
def sort_init_qpos():
    g = np.load(os.path.join(ROOT, "tests", "golden", "ref_offline_ik.npz"))
    return g["sorting__traj_last"].copy()


def _env(n, **kw):
    from d3il_amd.envs.sorting import SortingVecEnv
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    return SortingVecEnv(n, device=0, **kw)


def _oracle_state(o):
    """Oracle state in the device layout: arm q[9] v[9] | cubes (pos3 quat4 vel6) x 4."""
    qp, qv = o.state()
    cubes = np.concatenate([np.concatenate([qp[7 * b:7 * b + 7], qv[6 * b:6 * b + 6]]) for b in range(NB)])
    return np.concatenate([qp[7 * NB:7 * NB + 9], qv[6 * NB:6 * NB + 9]]), cubes


def _dev_err(st, e, o):
    arm, cubes = _oracle_state(o)
    d = st[42:42 + 13 * NB, e] - cubes
    vel = np.zeros(13 * NB, bool)
    for b in range(NB):
        vel[13 * b + 7:13 * b + 13] = True
    return max(np.abs(d[~vel]).max(), 1e-2 * np.abs(d[vel]).max(), np.abs(st[:9, e] - arm[:9]).max(), 1e-2 * np.abs(st[9:18, e] - arm[9:]).max())


def _action(des, z):
    n = des.shape[0]
    quat = torch.tensor([0.0, 1, 0, 0], dtype=torch.float64, device=des.device).expand(n, 4)
    return torch.cat([des, z, quat], dim=1).contiguous()


def test_reset_and_landing_match_oracle(sort_blob, sort_init_qpos):
    from oracle.oracle import Oracle
    from d3il_amd.envs.sorting import sample_contexts
    n = 40
    ctx = sample_contexts(n, NB, seed=3)
    env = _env(n)
    assert env.obs.shape == (n, 14) and env.state_rows == 42 + 13 * NB + 6 * NB + 9 + 2
    env.set_init_qpos(sort_init_qpos)
    obs = env.reset(context=ctx).cpu().numpy()
    st, fl, sc = env.get_state()
    assert (env.mode.cpu().numpy() == 240).all()
    check = [0, 7, 31, 32, 39]
    oracles = {}
    for e in check:
        o = Oracle(sort_blob)
        o.env_start(sort_init_qpos)
        oo = o.sort_reset(ctx[e].reshape(NB, 7))
        np.testing.assert_array_equal(obs[e], oo)
        assert _dev_err(st, e, o) < 1e-11 and sc[e] == 0 and not (fl[e] & BAD)
        oracles[e] = o
    z = env.robot_state()[:, 2:3].clone()
    des = env.obs[:, :2].to(torch.float64).clone()
    for t in range(12):
        a = _action(des, z)
        obs, rew, done, info = env.step(a)
        torch.cuda.synchronize()
        st, fl, sc = env.get_state()
        an = a.cpu().numpy()
        for e in check:
            oo, do, io = oracles[e].sort_step(an[e])
            np.testing.assert_array_equal(obs[e].cpu().numpy(), oo)
            assert _dev_err(st, e, oracles[e]) < 1e-10 and not (fl[e] & BAD)
            assert bool(done[e]) == do and int(info["mode"][e]) == io["mode"] == 240 and not bool(info["success"][e])
    assert float(rew.abs().max()) == 0.0
    env.close()


@pytest.mark.parametrize("fast", [1, 0])
def test_scripted_push_into_the_bin_matches_oracle(sort_blob, sort_init_qpos, fast):
    """Every environment pushes its red_1 cube over the platform edge into the red bin; six of them are followed by the oracle."""
    from oracle.oracle import Oracle
    from d3il_amd.envs.sorting import sample_contexts
    n = 64
    ctx = sample_contexts(n, NB, seed=11)
    env = _env(n)
    env.set_option("ik_fast_path", fast)
    env.set_init_qpos(sort_init_qpos)
    env.reset(context=ctx)
    check = [0, 13, 31, 32, 50, 63] if fast else [5, 40]
    oracles = {}
    for e in check:
        o = Oracle(sort_blob); o.env_start(sort_init_qpos); o.sort_reset(ctx[e].reshape(NB, 7)); oracles[e] = o
    z = env.robot_state()[:, 2:3].clone()
    des = env.obs[:, :2].to(torch.float64).clone()
    first_code, lost_at = {}, {}
    worst = 0.0
    for t in range(150):
        box = env.obs[:, 2:4].to(torch.float64)
        if t < 12:
            target = des.clone()
        else:
            aligned = ((des[:, 0] - box[:, 0]).abs() < 0.008) & (des[:, 1] < box[:, 1] - 0.02)
            target = torch.where(aligned[:, None], torch.stack([box[:, 0], torch.full_like(box[:, 0], 0.36)], 1), box + torch.tensor([0.0, -0.06], dtype=torch.float64, device=box.device))
        d = target - des
        nn = d.norm(dim=1, keepdim=True)
        des = des + d / nn.clamp_min(1e-9) * torch.minimum(nn, torch.full_like(nn, 0.006))
        a = _action(des, z)
        obs, rew, done, info = env.step(a)
        torch.cuda.synchronize()
        st, fl, sc = env.get_state()
        assert not (fl & BAD).any(), "solver failure / overflow / off-table flag at step %d" % t
        an = a.cpu().numpy()
        for e in list(oracles):
            oo, do, io = oracles[e].sort_step(an[e])
            err = _dev_err(st, e, oracles[e])
            if err > 1e-4:
                # the north star's bound (1e-4 on trajectory state) is the parity criterion.  A contact event (cube tipping over the
                # platform edge, cube meeting cube) amplifies the conditioning-level difference of two correct implementations
                # (tests/test_parity_conditioning.py): beyond this point the trajectories are different valid rollouts; the step at which
                # it happens is asserted below.  One-step parity through such events: test_one_step_parity_from_mid_episode_states.
                lost_at[e] = t
                del oracles[e]
                continue
            worst = max(worst, err)
            np.testing.assert_allclose(obs[e].cpu().numpy(), oo, rtol=1e-4, atol=1e-5)      # tan(yaw) entries grow without bound near 90 deg
            assert bool(done[e]) == do
            assert int(info["mode"][e]) == io["mode"] and bool(info["success"][e]) == io["success"]
            if io["mode"] != 240:
                first_code[e] = io["mode"]
                del oracles[e]          # parity shown up to the completion event; stop following this environment
    codes = env.mode.cpu().numpy()
    # every followed environment stays on the oracle's trajectory through reset transient, approach and the first pushes
    print("sorting scripted push: north-star (1e-4) horizon per followed env: %s (never exceeded: %s)" % (lost_at, sorted(set(check) - set(lost_at))))
    assert all(t >= 50 for t in lost_at.values()), (lost_at, worst)
    # and at least one of them all the way to the completion event, with the reference's mode code
    assert len(first_code) >= 1 and all(c == 0b01110000 for c in first_code.values()), (first_code, lost_at, worst)
    assert (codes == 0b01110000).sum() >= n // 3      # the simple script delivers most of the (randomly turned) red cubes
    env.close()


@pytest.mark.parametrize("strict", [0, 1])
def test_one_step_parity_from_mid_episode_states(sort_blob, sort_init_qpos, strict):
    """Random-walk set-points stir rod-cube, cube-cube and wall contacts; at several instants the oracle is loaded with the
    device state of a few environments and both take the same step.  All state rows incl. velocities, production stopping rule
    of the device solvers and the oracle's (solver_strict): the same bound holds - the difference is the conditioning of the
    soft-contact problem, not the solver tolerance (tests/test_parity_conditioning.py)."""
    from oracle.oracle import Oracle
    from d3il_amd.envs.sorting import sample_contexts
    n = 96
    ctx = sample_contexts(n, NB, seed=5)
    env = _env(n)
    env.set_option("solver_strict", strict)
    env.set_init_qpos(sort_init_qpos)
    env.reset(context=ctx)
    o = Oracle(sort_blob)
    o.env_start(sort_init_qpos)
    o.sort_reset(ctx[0].reshape(NB, 7))
    z = env.robot_state()[:, 2:3].clone()
    des = env.obs[:, :2].to(torch.float64).clone()
    g = torch.Generator(device="cpu").manual_seed(0)
    vel = torch.zeros(n, 2, dtype=torch.float64, device=des.device)
    worst = 0.0
    for t in range(90):
        box = env.obs[:, 2:14].to(torch.float64).reshape(n, NB, 3)[:, :, :2]
        tgt = box[torch.arange(n), (torch.arange(n) + t // 30) % NB]        # chase a cube, switch every 30 steps
        d = tgt - des
        nn = d.norm(dim=1, keepdim=True)
        vel = 0.7 * vel + 0.3 * (d / nn.clamp_min(1e-9) * 0.006 + 0.002 * torch.randn(n, 2, generator=g, dtype=torch.float64).to(des.device))
        des = des + vel
        a = _action(des, z)
        if t % 10 == 9:
            st, fl, sc = env.get_state()
            pick = [(t * 7 + k * 29) % n for k in range(3)]
        env.step(a)
        torch.cuda.synchronize()
        if t % 10 == 9:
            st2, fl2, sc2 = env.get_state()
            obs2 = env.obs.cpu().numpy()
            an = a.cpu().numpy()
            for e in pick:
                o.sort_set_state(st[:, e], int(fl[e]), int(sc[e]))
                oo, do, io = o.sort_step(an[e])
                err = _dev_err(st2, e, o)          # positions, and velocities weighted 1e-2: |dpos| < 2e-8, |dvel| < 2e-6 (north star 1e-4)
                worst = max(worst, err)
                assert err < 2e-8, (t, e, err)
                assert int(env.mode[e]) == io["mode"] and bool(env.success[e]) == io["success"]
        assert not (env.flags[:n].cpu().numpy() & ((1 << 16) | (1 << 18))).any()
    env.close()


def test_batch_size_and_lane_position_do_not_change_results(sort_init_qpos):
    """The same contexts in batches of 1, 17 (ragged last workgroup) and 80 environments evolve bit-identically, wherever they
    sit in a wave; a 8192-environment batch stays finite and unflagged."""
    from d3il_amd.envs.sorting import sample_contexts
    ctx = sample_contexts(80, NB, seed=21)
    finals = {}
    for n in (1, 17, 80):
        env = _env(n)
        env.set_init_qpos(sort_init_qpos)
        env.reset(context=ctx[:n])
        z = env.robot_state()[:, 2:3].clone()
        des = env.obs[:, :2].to(torch.float64).clone()
        for t in range(60):
            box = env.obs[:, 2:4].to(torch.float64)
            if t >= 12:
                aligned = ((des[:, 0] - box[:, 0]).abs() < 0.008) & (des[:, 1] < box[:, 1] - 0.02)
                target = torch.where(aligned[:, None], torch.stack([box[:, 0], torch.full_like(box[:, 0], 0.36)], 1), box + torch.tensor([0.0, -0.06], dtype=torch.float64, device=box.device))
                d = target - des
                nn = d.norm(dim=1, keepdim=True)
                des = des + d / nn.clamp_min(1e-9) * torch.minimum(nn, torch.full_like(nn, 0.006))
            env.step(_action(des, z))
        torch.cuda.synchronize()
        st, fl, sc = env.get_state()
        finals[n] = st
        assert not (fl & BAD).any()
        env.close()
    assert np.array_equal(finals[1][:, 0], finals[80][:, 0]) and np.array_equal(finals[17], finals[80][:, :17])
    big = _env(8192)
    big.set_init_qpos(sort_init_qpos)
    big.reset(context=np.tile(ctx, (103, 1))[:8192])
    z = big.robot_state()[:, 2:3].clone()
    des = big.obs[:, :2].to(torch.float64).clone()
    for t in range(20):
        big.step(_action(des, z))
    torch.cuda.synchronize()
    st, fl, sc = big.get_state()
    assert np.isfinite(st).all() and not (fl & BAD).any() and (sc == 20).all()
    # tiles of the 80 contexts are bit-identical across the batch
    assert np.array_equal(st[:, :80], st[:, 80:160]) and np.array_equal(st[:, :80], st[:, 8000:8080])
    big.close()


def test_sorting_2_reset_landing_and_push_match_oracle(sort_init_qpos):
    """The two-box scene through the same kernels (two live lanes per group): oracle parity through reset, landing, approach and
    the first pushes; protocol shapes of the num_boxes = 2 env."""
    from oracle.oracle import Oracle
    from d3il_amd.envs.sorting import SortingVecEnv, sample_contexts
    from d3il_amd.model import blob
    n = 24
    ctx = sample_contexts(n, 2, seed=8)
    env = SortingVecEnv(n, device=0, num_boxes=2)
    assert env.obs.shape == (n, 8) and env.state_rows == 42 + 26 + 21 + 2
    env.set_init_qpos(sort_init_qpos)
    env.reset(context=ctx)
    assert (env.mode.cpu().numpy() == 0b11000000).all()
    m2 = blob.load("sorting_2")
    check = [0, 7, 16, 23]
    oracles = {}
    for e in check:
        o = Oracle(m2); o.env_start(sort_init_qpos); o.sort_reset(ctx[e].reshape(2, 7)); oracles[e] = o
    z = env.robot_state()[:, 2:3].clone()
    des = env.obs[:, :2].to(torch.float64).clone()
    for t in range(45):
        box = env.obs[:, 2:4].to(torch.float64)
        if t >= 12:
            aligned = ((des[:, 0] - box[:, 0]).abs() < 0.008) & (des[:, 1] < box[:, 1] - 0.02)
            target = torch.where(aligned[:, None], torch.stack([box[:, 0], torch.full_like(box[:, 0], 0.36)], 1), box + torch.tensor([0.0, -0.06], dtype=torch.float64, device=box.device))
            d = target - des
            nn = d.norm(dim=1, keepdim=True)
            des = des + d / nn.clamp_min(1e-9) * torch.minimum(nn, torch.full_like(nn, 0.006))
        a = _action(des, z)
        obs, rew, done, info = env.step(a)
        torch.cuda.synchronize()
        st, fl, sc = env.get_state()
        assert not (fl & BAD).any()
        an = a.cpu().numpy()
        for e in check:
            oo, do, io = oracles[e].sort_step(an[e])
            qp, qv = oracles[e].state()
            cubes = np.concatenate([np.concatenate([qp[7 * b:7 * b + 7], qv[6 * b:6 * b + 6]]) for b in range(2)])
            d = st[42:68, e] - cubes
            pos = np.array([k % 13 < 7 for k in range(26)])
            assert np.abs(d[pos]).max() < 1e-6 and np.abs(st[:9, e] - qp[14:23]).max() < 1e-6, (t, e)
            np.testing.assert_allclose(obs[e].cpu().numpy(), oo, rtol=1e-4, atol=1e-5)
            assert bool(done[e]) == do and int(info["mode"][e]) == io["mode"]
    env.close()


def test_cube_at_the_table_edge_meets_the_frame_beams(sort_blob, sort_init_qpos):
    """VERDICT r3 missing #3: the cube <-> frame-beam pairs (lab_surrounding.xml:3-114) on the device.  Cubes are dropped along the front and
    the left edge of the table - centre inside the old 8 cm margin, on the edge, 1 cm beyond it (over the beam) - and followed by the oracle,
    which has always evaluated these pairs: same trajectory while the beams carry a cube (the device used to raise OFF_TABLE there and
    ignore the beams), OFF_TABLE only once a cube centre has passed the outer face of the frame."""
    from oracle.oracle import Oracle
    from tests.test_sorting_oracle import CTX
    spots = [(0.86, 0.35), (0.885, 0.35), (0.90, 0.35), (0.895, -0.6), (0.45, 0.955), (0.45, 0.985), (0.30, -0.99), (0.905, 0.97)]
    n = len(spots)
    ctx = np.tile(CTX.reshape(1, NB, 7), (n, 1, 1))
    for e, (x, y) in enumerate(spots):
        ctx[e, 0, :2] = [x, y]
        ctx[e, 3, :2] = [x - 0.2 if x > 0.5 else x + 0.1, y * 0.5]      # a second cube of the environment well inside the table
    env = _env(n)
    env.set_init_qpos(sort_init_qpos)
    env.reset(context=ctx.reshape(n, NB * 7))
    torch.cuda.synchronize()
    orc = []
    for e in range(n):
        o = Oracle(sort_blob)
        o.env_start(sort_init_qpos)
        o.sort_reset(ctx[e])
        orc.append(o)
    z = env.robot_state()[:, 2:3].clone()
    des = env.obs[:, :2].to(torch.float64).clone()
    beam_steps, worst = 0, 0.0
    outer_x, outer_y = 0.91, 1.0
    for t in range(14):
        a = _action(des, z)
        env.step(a)
        torch.cuda.synchronize()
        st, fl, sc = env.get_state()
        an = a.cpu().numpy()
        for e in range(n):
            orc[e].sort_step(an[e])
            con = orc[e].contacts()
            on_beam = any(int(row[8]) in range(2, 16) for row in con)            # geoms 2 .. 15: the aluminium profiles
            beam_steps += on_beam
            err = _dev_err(st, e, orc[e])
            worst = max(worst, err)
            assert err < 1e-7, (t, e, err)
            x, y = st[42, e], st[43, e]
            off = bool(fl[e] & (1 << 19))
            assert (off or (x <= outer_x and abs(y) <= outer_y)) and (not off or x > outer_x - 0.01 or abs(y) > outer_y - 0.01), (t, e, x, y, off)
            assert not (fl[e] & ((1 << 16) | (1 << 18)))
    print("cubes on the frame beams: %d environment-steps, max |device - oracle| %.2e" % (beam_steps, worst))
    assert beam_steps >= 10
    assert not (env.flags[:1].cpu().numpy() & (1 << 19)).any()      # 3 cm inside the edge: no flag (it was one before)
    env.close()
