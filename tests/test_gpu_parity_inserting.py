"""Inserting (Gate_Insertion_Env): the HIP path through the C ABI against the CPU oracle.

The task runs on the generic engine of the Sorting task with three cubes, the seventeen static walls of the gates and - new for this
task - the rod <-> wall contacts.  As for Pushing / Sorting, contact-rich pushing separates two f64 implementations over long horizons
(the rod chattering on a wall makes the contact set itself depend on the last digits), so parity is asserted at reset, along scripted
pushes that drive a cube and then the rod into the walls (bounded tolerance, horizon asserted), and as ONE-STEP parity from mid-episode
device states incl. states with the rod on a wall.  Integer outputs (done, success, mode code, number of letters) are compared exactly;
the success / mode logic is driven through its events by loading states with the cubes on their goals.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BAD = (1 << 16) | (1 << 18) | (1 << 19)          # solver fail, contact overflow, off table
NB = 3


@pytest.fixture(scope="module")
def ins_blob():
    from d3il_amd.model import blob
    return blob.load("inserting")


@pytest.fixture(scope="module")
def ins_init_qpos():
    # init_end_eff_pos of the task is the Avoiding / Pushing start pose (gate_insertion_objects.py:5): same offline-IK result
    g = np.load(os.path.join(ROOT, "tests", "golden", "ref_offline_ik.npz"))
    return g["avoiding__traj_last"].copy()


def _env(n, **kw):
    from d3il_amd.envs.inserting import GateInsertionVecEnv
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    return GateInsertionVecEnv(n, device=0, **kw)


def _oracle_state(o):
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


def _rod_on_static(o, blob):
    """The oracle's current contact list holds a rod <-> static box pair (the rod on a wall of the gates)."""
    con = o.contacts()
    if not len(con):
        return False
    cubes = {g for g in range(blob.ngeom) if blob.geom_body[g] in [blob.obj_body[k] for k in range(blob.n_obj)]}
    return any(int(b) == blob.rod_geom and int(a) not in cubes for a, b in con[:, 8:10])


def _script(des, obs, t, k):
    """Way-points: behind cube k (south of it), then north through it towards the gates."""
    box = obs[:, 2 + 3 * k:4 + 3 * k].to(torch.float64)
    if t < 8:
        return des.clone()
    aligned = ((des[:, 0] - box[:, 0]).abs() < 0.008) & (des[:, 1] < box[:, 1] - 0.02)
    north = torch.stack([box[:, 0], torch.full_like(box[:, 0], 0.30)], 1)
    behind = box + torch.tensor([0.0, -0.055], dtype=torch.float64, device=box.device)
    return torch.where(aligned[:, None], north, behind)


def _targets(des, obs, t):
    """Environment e pushes cube e % 3 north; every fourth environment instead runs the rod itself into the walls of the left gate (whatever lies on
    the way is pushed along): to (0.45, 0.10), then towards (0.40, 0.26) - the diagonal wall maze_3 and the long wall maze_5 stand there."""
    n = des.shape[0]
    idx = torch.arange(n)
    tgt = torch.stack([_script(des, obs, t, k) for k in range(NB)], 0)[idx % NB, idx]
    wall = torch.tensor([0.45, 0.10] if t < 70 else [0.40, 0.26], dtype=torch.float64, device=des.device).expand(n, 2)
    if t < 8:
        wall = des
    return torch.where((idx % 4 == 3).to(des.device)[:, None], wall, tgt)


def test_reset_matches_oracle_and_protocol(ins_blob, ins_init_qpos):
    from oracle.oracle import Oracle
    from d3il_amd.envs.inserting import sample_contexts
    n = 40
    ctx = sample_contexts(n, seed=3)
    env = _env(n)
    assert env.obs.shape == (n, 11) and env.state_rows == 110 and env.max_steps_per_episode == 2000
    env.set_init_qpos(ins_init_qpos)
    obs = env.reset(context=ctx).cpu().numpy()
    st, fl, sc = env.get_state()
    assert (env.mode.cpu().numpy() == 0).all()
    check = [0, 7, 31, 32, 39]
    oracles = {}
    for e in check:
        o = Oracle(ins_blob)
        o.env_start(ins_init_qpos)
        oo = o.ins_reset(ctx[e].reshape(NB, 7))
        np.testing.assert_array_equal(obs[e], oo)
        assert _dev_err(st, e, o) < 1e-11 and sc[e] == 0 and not (fl[e] & BAD)
        oracles[e] = o
    z = env.robot_state()[:, 2:3].clone()
    des = env.obs[:, :2].to(torch.float64).clone()
    for t in range(10):                                # the cubes settle on the table (placed at z = 0: inside it, pushed out)
        a = _action(des, z)
        obs, rew, done, info = env.step(a)
        torch.cuda.synchronize()
        st, fl, sc = env.get_state()
        an = a.cpu().numpy()
        for e in check:
            oo, do, io = oracles[e].ins_step(an[e])
            np.testing.assert_array_equal(obs[e].cpu().numpy(), oo)
            assert _dev_err(st, e, oracles[e]) < 1e-9 and not (fl[e] & BAD)      # the settling transient: 1.6e-10 observed
            assert bool(done[e]) == do and int(info["mode"][e]) == io["mode"] == 0 and not bool(info["success"][e])
            assert abs(float(info["mean_distance"][e]) - io["mean_distance"]) < 1e-10 and int(info["one_box_success"][e]) == 0
    # reward = -(min robot <-> box distance in xy + the three box <-> target distances), gate_insertion.py:448-473
    pos, _ = env.box_state()
    tg = np.array([[0.3575, 0.276, 0.0], [0.525, 0.4535, 0.0], [0.6925, 0.276, 0.0]])
    p = pos.cpu().numpy()
    tcp = env.robot_state().cpu().numpy()
    want = -(np.linalg.norm(p[:, :, :2] - tcp[:, None, :2], axis=2).min(1) + np.linalg.norm(p - tg[None], axis=2).sum(1))
    np.testing.assert_allclose(env.get_reward().cpu().numpy(), want, atol=1e-12)
    # the reward a step returns is sampled BEFORE its physics, like obs / done (gym_env_wrapper.py:88-93)
    des = des + 0.004
    pre = env.get_reward().clone()
    _, rew, _, _ = env.step(_action(des, z))
    torch.cuda.synchronize()
    assert torch.equal(rew, pre) and not torch.equal(env.get_reward(), pre)
    env.close()


@pytest.mark.parametrize("fast", [1, 0])
def test_scripted_pushes_into_the_gates_match_oracle(ins_blob, ins_init_qpos, fast):
    """Every environment drives the rod behind one of its cubes and north through it: the cube meets the walls of the gates, then the rod does.
    Six environments are followed by the oracle."""
    from oracle.oracle import Oracle
    from d3il_amd.envs.inserting import sample_contexts
    n = 64
    ctx = sample_contexts(n, seed=11)
    env = _env(n)
    env.set_option("ik_fast_path", fast)
    env.set_init_qpos(ins_init_qpos)
    env.reset(context=ctx)
    check = [0, 13, 31, 32, 50, 63] if fast else [5, 40]
    oracles = {}
    for e in check:
        o = Oracle(ins_blob); o.env_start(ins_init_qpos); o.ins_reset(ctx[e].reshape(NB, 7)); oracles[e] = o
    z = env.robot_state()[:, 2:3].clone()
    des = env.obs[:, :2].to(torch.float64).clone()
    which = torch.arange(n) % NB
    lost_at, rod_wall_steps, worst = {}, 0, 0.0
    for t in range(150):
        tgt = _targets(des, env.obs, t)
        d = tgt - des
        nn = d.norm(dim=1, keepdim=True)
        des = des + d / nn.clamp_min(1e-9) * torch.minimum(nn, torch.full_like(nn, 0.006))
        a = _action(des, z)
        obs, rew, done, info = env.step(a)
        torch.cuda.synchronize()
        st, fl, sc = env.get_state()
        assert not (fl & BAD).any(), "solver failure / overflow / off-table flag at step %d" % t
        an = a.cpu().numpy()
        for e in list(oracles):
            oo, do, io = oracles[e].ins_step(an[e])
            err = _dev_err(st, e, oracles[e])
            if err > 1e-4:          # the north star's bound; beyond it the two rollouts are different valid ones (cf. the Sorting test)
                lost_at[e] = t
                del oracles[e]
                continue
            worst = max(worst, err)
            rod_wall_steps += int(_rod_on_static(oracles[e], ins_blob))
            np.testing.assert_allclose(obs[e].cpu().numpy(), oo, rtol=1e-3, atol=1e-5)      # tan(yaw) entries amplify the state difference by 1 + tan^2
            assert bool(done[e]) == do and int(info["mode"][e]) == io["mode"] and bool(info["success"][e]) == io["success"]
            assert int(env.mode[e]) >> 3 == io["n_mode"] and abs(float(info["mean_distance"][e]) - io["mean_distance"]) < 1e-5
    print("inserting scripted push: north-star (1e-4) horizon per followed env: %s; steps with the rod on a wall while followed: %d; worst %.2e" % (lost_at, rod_wall_steps, worst))
    assert all(t >= 60 for t in lost_at.values()), (lost_at, worst)
    if fast:
        assert rod_wall_steps >= 1, "no followed environment reached a rod <-> wall contact"
    # the cubes were really moved: in most environments the pushed cube has left its start position by centimetres
    moved = (env.obs[:, 2:].reshape(n, NB, 3)[torch.arange(n), which, :2].cpu().numpy() - ctx.reshape(n, NB, 7)[np.arange(n), which.numpy(), :2])
    assert (np.linalg.norm(moved, axis=1) > 0.03).sum() >= n // 3
    env.close()


@pytest.mark.parametrize("strict", [0, 1])
def test_one_step_parity_from_mid_episode_states(ins_blob, ins_init_qpos, strict):
    """Scripted pushes stir rod-cube, cube-wall, cube-cube and rod-wall contacts; at several instants the oracle is loaded with the device state of a
    few environments and both take the same step (all state rows incl. velocities; production and oracle-like stopping rule)."""
    from oracle.oracle import Oracle
    from d3il_amd.envs.inserting import sample_contexts
    n = 96
    ctx = sample_contexts(n, seed=5)
    env = _env(n)
    env.set_option("solver_strict", strict)
    env.set_init_qpos(ins_init_qpos)
    env.reset(context=ctx)
    o = Oracle(ins_blob)
    o.env_start(ins_init_qpos)
    o.ins_reset(ctx[0].reshape(NB, 7))
    z = env.robot_state()[:, 2:3].clone()
    des = env.obs[:, :2].to(torch.float64).clone()
    which = torch.arange(n) % NB
    worst, rod_wall, with_contacts = 0.0, 0, 0
    for t in range(140):
        tgt = _targets(des, env.obs, t)
        d = tgt - des
        nn = d.norm(dim=1, keepdim=True)
        des = des + d / nn.clamp_min(1e-9) * torch.minimum(nn, torch.full_like(nn, 0.006))
        a = _action(des, z)
        probe = (t % 10 == 9 and t > 40) or (t >= 84 and t % 3 == 0)
        if probe:
            st, fl, sc = env.get_state()
            pick = [(t * 7 + k * 29) % n for k in range(4)] if t % 10 == 9 else []
            pick += [3 + 4 * (t % 24), 3 + 4 * ((5 * t + 7) % 24)]      # wall runners: from step ~84 on their rods work against maze_3 / maze_5
        env.step(a)
        torch.cuda.synchronize()
        if probe:
            st2, fl2, sc2 = env.get_state()
            an = a.cpu().numpy()
            for e in pick:
                o.ins_set_state(st[:, e], int(fl[e]), int(sc[e]))
                oo, do, io = o.ins_step(an[e])
                err = _dev_err(st2, e, o)          # positions, and velocities weighted 1e-2: |dpos| < 2e-8, |dvel| < 2e-6 (north star 1e-4)
                worst = max(worst, err)
                rod_wall += int(_rod_on_static(o, ins_blob))
                with_contacts += int(len(o.contacts()) > 12)
                assert err < 2e-8, (t, e, err)
                assert int(env.mode[e]) == (io["mode"] | (io["n_mode"] << 3)) and bool(env.success[e]) == io["success"]
                assert abs(float(env.state[env.task_row + 1, e]) - io["mean_distance"]) < 1e-9
        assert not (env.flags[:n].cpu().numpy() & ((1 << 16) | (1 << 18))).any()
    print("inserting one-step parity: worst %.2e, probes with the rod on a wall: %d, probes with pushing contacts: %d" % (worst, rod_wall, with_contacts))
    assert rod_wall >= 5 and with_contacts >= 10
    env.close()


def test_success_and_mode_events_from_loaded_states(ins_blob, ins_init_qpos):
    """The cubes are put on their goals one after the other (d3il_set_state), in a different order per environment: number of letters, mode code
    (mode_dict), success, done of the NEXT step and the n-box successes follow the oracle through every event."""
    import itertools
    from oracle.oracle import Oracle
    from d3il_amd.envs.inserting import MODE_DICT, sample_contexts
    orders = list(itertools.permutations(range(3)))
    n = len(orders)
    ctx = sample_contexts(n, seed=2)
    env = _env(n)
    env.set_init_qpos(ins_init_qpos)
    env.reset(context=ctx)
    z = env.robot_state()[:, 2:3].clone()
    des = env.obs[:, :2].to(torch.float64).clone()
    a = _action(des, z)
    for t in range(4):
        env.step(a)
    torch.cuda.synchronize()
    tg = np.array([[0.3575, 0.276, 0.0], [0.525, 0.4535, 0.0], [0.6925, 0.276, 0.0]])
    oracles = []
    for e in range(n):
        o = Oracle(ins_blob); o.env_start(ins_init_qpos); o.ins_reset(ctx[e].reshape(NB, 7)); oracles.append(o)
    an = a.cpu().numpy()
    for stage in range(3):
        st, fl, sc = env.get_state()
        rest_z = st[42 + 2, 0]
        for e in range(n):
            k = orders[e][stage]
            r = 42 + 13 * k
            st[r:r + 3, e] = [tg[k][0] + 0.002 * (e % 2), tg[k][1] - 0.001, rest_z]
            st[r + 3:r + 7, e] = [1, 0, 0, 0]
            st[r + 7:r + 13, e] = 0
        env.set_state(st, fl, sc)
        for e in range(n):
            oracles[e].ins_set_state(st[:, e], int(fl[e]), int(sc[e]))
        for t in range(2):
            obs, rew, done, info = env.step(a)
            torch.cuda.synchronize()
            for e in range(n):
                oo, do, io = oracles[e].ins_step(an[e])
                np.testing.assert_array_equal(obs[e].cpu().numpy(), oo)
                assert bool(done[e]) == do and int(info["mode"][e]) == io["mode"] and bool(info["success"][e]) == io["success"], (stage, t, e)
                assert int(env.mode[e]) >> 3 == io["n_mode"] == stage + 1
                assert [int(info[k][e]) for k in ("one_box_success", "two_box_success", "three_box_success")] == [1, int(stage >= 1), int(stage >= 2)]
    letters = env.mode_letters()
    for e in range(n):
        want = "".join("rgb"[k] for k in orders[e])
        assert letters[e] == want and int(env.mode[e]) & 7 == MODE_DICT[want] and bool(env.success[e]) and bool(env.done[e])
    env.close()


def test_batch_size_and_lane_position_do_not_change_results(ins_init_qpos):
    """The same contexts in batches of 1, 17 (ragged last workgroup) and 80 environments evolve bit-identically wherever they sit in a wave; a masked
    reset touches only the masked environments; an 8192-environment batch stays finite and unflagged."""
    from d3il_amd.envs.inserting import sample_contexts
    ctx = sample_contexts(80, seed=21)
    finals = {}
    for n in (1, 17, 80):
        env = _env(n)
        env.set_init_qpos(ins_init_qpos)
        env.reset(context=ctx[:n])
        z = env.robot_state()[:, 2:3].clone()
        des = env.obs[:, :2].to(torch.float64).clone()
        which = torch.arange(n) % NB
        for t in range(90):
            tgt = _targets(des, env.obs, t)
            d = tgt - des
            nn = d.norm(dim=1, keepdim=True)
            des = des + d / nn.clamp_min(1e-9) * torch.minimum(nn, torch.full_like(nn, 0.006))
            env.step(_action(des, z))
        torch.cuda.synchronize()
        st, fl, sc = env.get_state()
        finals[n] = st
        assert not (fl & BAD).any()
        if n == 80:
            mask = torch.zeros(n, dtype=torch.uint8, device=env.device)
            mask[[3, 40, 79]] = 1
            env.reset(mask=mask, context=ctx)
            torch.cuda.synchronize()
            st2, fl2, sc2 = env.get_state()
            keep = np.ones(n, bool); keep[[3, 40, 79]] = False
            assert np.array_equal(st2[:, keep], st[:, keep]) and (sc2[~keep] == 0).all() and (sc2[keep] == 90).all()
        env.close()
    assert np.array_equal(finals[1][:, 0], finals[80][:, 0]) and np.array_equal(finals[17], finals[80][:, :17])
    big = _env(8192)
    big.set_init_qpos(ins_init_qpos)
    big.reset(context=np.tile(ctx, (103, 1))[:8192])
    z = big.robot_state()[:, 2:3].clone()
    des = big.obs[:, :2].to(torch.float64).clone()
    for t in range(20):
        big.step(_action(des, z))
    torch.cuda.synchronize()
    st, fl, sc = big.get_state()
    assert np.isfinite(st).all() and not (fl & BAD).any() and (sc == 20).all()
    assert np.array_equal(st[:, :80], st[:, 80:160]) and np.array_equal(st[:, :80], st[:, 8000:8080])
    big.close()


def test_physically_produced_insertions_match_oracle(ins_blob, ins_init_qpos):
    """Mode events produced by the PHYSICS (not by loaded states): the red cube starts in the mouth of its gate (x 0.41 .. 0.425, small y / yaw variations), the
    rod comes up east of it and pushes it west between maze_5 / maze_13 until maze_9 stops it 2 mm from its goal - the env's `modes` list gets its 'r'.
    One environment does the same with the green cube from (0.525, 0.36) north into its gate.  Four environments are followed by the oracle step by step:
    same number of letters at every step, the letter appears at the same step, same final code; most of the batch inserts."""
    from oracle.oracle import Oracle
    n = 16
    rng = np.random.default_rng(4)
    park = np.array([[0.40, -0.17, 0, 1, 0, 0, 0], [0.62, -0.08, 0, 1, 0, 0, 0], [0.45, 0.02, 0, 1, 0, 0, 0]], float)
    ctx = np.tile(park[None], (n, 1, 1))
    for e in range(n - 1):
        yaw = rng.uniform(-0.03, 0.03)
        ctx[e, 0, :2] = [0.41 + 0.001 * e, 0.276 + rng.uniform(-0.001, 0.001)]
        ctx[e, 0, 3:] = [np.cos(yaw / 2), 0, 0, np.sin(yaw / 2)]
    ctx[n - 1, 1, :2] = [0.525, 0.36]
    ctx[n - 1, 2, :2] = [0.40, 0.02]
    way = np.zeros((n, 3, 2))
    way[:] = np.array([(0.49, 0.10), (0.49, 0.276), (0.388, 0.276)])
    way[n - 1] = np.array([(0.525, 0.30), (0.525, 0.425), (0.525, 0.425)])
    env = _env(n)
    env.set_init_qpos(ins_init_qpos)
    env.reset(context=ctx.reshape(n, 21))
    check = [0, 7, 14, 15]
    oracles = {}
    for e in check:
        o = Oracle(ins_blob); o.env_start(ins_init_qpos); o.ins_reset(ctx[e]); oracles[e] = o
    z = env.robot_state()[:, 2:3].clone()
    des = env.obs[:, :2].to(torch.float64).clone()
    wayt = torch.as_tensor(way, dtype=torch.float64, device=des.device)
    wi = torch.zeros(n, dtype=torch.long, device=des.device)
    ar = torch.arange(n, device=des.device)
    first_dev, first_orc = {}, {}
    for t in range(170):
        d = wayt[ar, wi] - des
        nn = d.norm(dim=1, keepdim=True)
        wi = torch.where((nn[:, 0] < 1e-9) & (wi < 2), wi + 1, wi)
        d = wayt[ar, wi] - des
        nn = d.norm(dim=1, keepdim=True)
        des = des + d / nn.clamp_min(1e-12) * torch.minimum(nn, torch.full_like(nn, 0.006))
        a = _action(des, z)
        obs, rew, done, info = env.step(a)
        torch.cuda.synchronize()
        assert not (env.flags[:n].cpu().numpy() & BAD).any()
        nm = (env.mode.to(torch.int32) >> 3).cpu().numpy()
        an = a.cpu().numpy()
        for e in range(n):
            if nm[e] and e not in first_dev:
                first_dev[e] = t
        for e in check:
            oo, do, io = oracles[e].ins_step(an[e])
            if io["n_mode"] and e not in first_orc:
                first_orc[e] = t
            assert nm[e] == io["n_mode"] and bool(done[e]) == do and bool(info["success"][e]) == io["success"], (t, e, nm[e], io)
    letters = env.mode_letters()
    print("inserting, physical insertions: first letter at step (device) %s, (oracle) %s, letters %s" % (first_dev, first_orc, letters))
    assert {e: first_dev.get(e) for e in check} == {e: first_orc.get(e) for e in check}
    assert sum(1 for e in range(n - 1) if letters[e] == "r") >= 10 and all(l in ("", "r") for l in letters[:n - 1]) and letters[n - 1] in ("", "g")
    assert any(e in first_orc for e in check)
    env.close()
