"""Aligning: the HIP path through the C ABI against the CPU oracle.

The device runs the task on variant 2 of the wave-cooperative engine (d3il_amd/csrc/align_step.h: one free compound body of five box geoms in
centre-of-mass coordinates, rod <-> geom cylinder-box jobs); the oracle is the generic MuJoCo-style engine of oracle/d3il_oracle.c, whose task logic
is pinned to the reference's own Python (tests/test_aligning_oracle.py).  Asserted: reset on all 60 reference contexts, scripted pushes from inside
and from outside the walls followed step by step, ONE-STEP parity from mid-episode device states on all state rows, exact done / success / mode,
batch properties, and the Sim class' integer tables against whole oracle episodes.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BAD = (1 << 16) | (1 << 18) | (1 << 19) | (1 << 20)      # solver fail, contact overflow, off table, hand near


@pytest.fixture(scope="module")
def align_blob():
    from d3il_amd.model import blob
    return blob.load("aligning")


@pytest.fixture(scope="module")
def ctx60():
    from d3il_amd.envs.aligning import load_test_contexts
    return load_test_contexts()


def _env(n, **kw):
    from d3il_amd.envs.aligning import RobotPushVecEnv
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    return RobotPushVecEnv(n, device=0, **kw)


def _err(st, e, o):
    """max |device - oracle| over positions (arm q, box pos / quat) and over velocities."""
    arm, box = o.align_state()
    dp = max(np.abs(st[0:9, e] - arm[:9]).max(), np.abs(st[42:49, e] - box[:7]).max())
    dv = max(np.abs(st[9:18, e] - arm[9:]).max(), np.abs(st[49:55, e] - box[7:]).max())
    return dp, dv


def _action(des):
    n = des.shape[0]
    quat = torch.tensor([0.0, 1, 0, 0], dtype=torch.float64, device=des.device).expand(n, 4)
    return torch.cat([des, quat], dim=1).contiguous()


def _toward(des, target, step):
    d = target - des
    nn = d.norm(dim=1, keepdim=True).clamp_min(1e-12)
    return des + d / nn * torch.minimum(nn, torch.full_like(nn, step))


def test_reset_matches_oracle_on_all_reference_contexts(align_blob, ctx60):
    from oracle.oracle import Oracle
    n = 60
    env = _env(n)
    assert env.obs.shape == (n, 17) and env.state_rows == 77
    q0 = env.start()[0]
    obs = env.reset(context=ctx60).cpu().numpy()
    st, fl, sc = env.get_state()
    o = Oracle(align_blob)
    o.env_start(q0)
    worst = 0.0
    for e in range(n):
        oo = o.align_reset(ctx60[e])
        np.testing.assert_array_equal(obs[e], oo)
        dp, dv = _err(st, e, o)
        worst = max(worst, dp, 1e-2 * dv)
        assert dp < 1e-10 and dv < 1e-8 and sc[e] == 0 and not (fl[e] & BAD) and not (fl[e] & (1 << 15))
        np.testing.assert_array_equal(st[70:77, e], ctx60[e, 7:14])
    assert (env.mode.cpu().numpy() == -1).all() and not env.done.any()
    print("aligning reset: max |device - oracle| %.2e" % worst)
    env.close()


def _script(ctx, t, des):
    """Two scripted pushes: even environments go down between the walls and push the far wall from inside (mode 0); odd ones go down outside,
    off-centre, and push the box from outside so that it turns (mode 1) - the two episodes of tests/test_aligning_oracle.py."""
    n = des.shape[0]
    cx, cy = ctx[:, 0:1], ctx[:, 1:2]
    inside = (torch.arange(n, device=des.device) % 2 == 0).unsqueeze(1)
    if t < 40:
        tgt_in = torch.cat([cx, cy, torch.full_like(cx, 0.25)], 1); tgt_out = torch.cat([cx + 0.03, cy - 0.12, torch.full_like(cx, 0.25)], 1); step = 0.008
    elif t < 80:
        tgt_in = torch.cat([cx, cy, torch.full_like(cx, 0.14)], 1); tgt_out = torch.cat([cx + 0.03, cy - 0.12, torch.full_like(cx, 0.14)], 1); step = 0.008
    else:
        tgt_in = torch.cat([cx, cy + 0.12, torch.full_like(cx, 0.14)], 1); tgt_out = torch.cat([cx + 0.03, cy + 0.05, torch.full_like(cx, 0.14)], 1); step = 0.004
    return _toward(des, torch.where(inside, tgt_in, tgt_out), step)


def test_scripted_pushes_follow_the_oracle(align_blob, ctx60):
    """Approach, descent and 50 steps of pushing - from inside the walls (the rod drags the box: mode 0) and from outside, off-centre (the box turns:
    mode 1): positions within 1e-6 of the oracle, observations within f32 round-off of them, done / success / mode exact."""
    from oracle.oracle import Oracle
    n = 16
    env = _env(n)
    q0 = env.start()[0]
    ctx = ctx60[np.arange(n) * 3 % 60]
    env.reset(context=ctx)
    ctx_t = torch.as_tensor(ctx, device=env.device)
    follow = [0, 1, 6, 7, 12, 13]
    orc = {}
    for e in follow:
        o = Oracle(align_blob)
        o.env_start(q0)
        o.align_reset(ctx[e])
        orc[e] = o
    des = env.robot_state().clone()
    worst, modes = 0.0, set()
    for t in range(130):
        des = _script(ctx_t, t, des)
        a = _action(des)
        obs, rew, done, info = env.step(a)
        torch.cuda.synchronize()
        st, fl, sc = env.get_state()
        an = a.cpu().numpy()
        for e in follow:
            oo, ro, do, io = orc[e].align_step(an[e])
            dp, dv = _err(st, e, orc[e])
            worst = max(worst, dp)
            assert dp < 1e-6 and dv < 1e-3, (t, e, dp, dv)
            assert np.abs(obs[e].cpu().numpy() - oo).max() < 2e-6 and bool(done[e]) == do
            assert int(info["mode"][e]) == io["mode"] and bool(info["success"][e]) == io["success"]
            assert abs(float(info["mean_distance"][e]) - io["mean_distance"]) < 1e-5 and abs(float(rew[e]) - ro) < 1e-5
            assert not (fl[e] & BAD), hex(fl[e])
            modes.add(io["mode"])
    bp, bq = env.box_state()
    moved = (bp[:, :2] - ctx_t[:, :2]).norm(dim=1).cpu().numpy()
    assert modes == {0, 1} and moved[::2].min() > 0.03 and moved[1::2].min() > 0.01
    print("aligning scripted pushes: max |dpos| over 130 steps %.2e; box displacement %.3f .. %.3f m" % (worst, moved.min(), moved.max()))
    env.close()


def test_one_step_parity_from_mid_episode_states(align_blob, ctx60):
    """Device states sampled along the scripted pushes (rest, rod against a wall from inside / outside, the box turning) are loaded into the oracle;
    one env step (35 sub-steps) with the same action agrees on ALL state rows: |dpos| <= 2e-8, |dvel| <= 2e-6 (the bounds of the Pushing twin),
    exact done / success / mode."""
    from oracle.oracle import Oracle
    n = 64
    env = _env(n)
    q0 = env.start()[0]
    ctx = ctx60[np.arange(n) % 60]
    env.reset(context=ctx)
    ctx_t = torch.as_tensor(ctx, device=env.device)
    o = Oracle(align_blob)
    o.env_start(q0)
    des = env.robot_state().clone()
    rng = np.random.default_rng(1)
    checked = in_contact = 0
    wp = wv = 0.0
    for t in range(125):
        des = _script(ctx_t, t, des)
        a = _action(des)
        torch.cuda.synchronize()
        st0, fl0, sc0 = env.get_state()
        obs, rew, done, info = env.step(a)
        torch.cuda.synchronize()
        st1, fl1, sc1 = env.get_state()
        if t < 70 or (t < 84 and t % 3) or (t >= 84 and t % 2):
            continue
        an = a.cpu().numpy()
        for e in rng.choice(n, 5, replace=False):
            o.align_set_state(st0[:, e], step=sc0[e], terminated=bool(fl0[e] & (1 << 12)), ik_valid=bool(fl0[e] & (1 << 15)))
            oo, ro, do, io = o.align_step(an[e])
            dp, dv = _err(st1, e, o)
            wp, wv = max(wp, dp), max(wv, dv)
            assert dp < 2e-8 and dv < 2e-6, (t, e, dp, dv)
            assert bool(done[e]) == do and int(info["mode"][e]) == io["mode"] and bool(info["success"][e]) == io["success"]
            assert not (fl1[e] & BAD), hex(fl1[e])
            checked += 1
            in_contact += len(o.contacts()) > 4
    print("aligning one-step parity: %d states (%d with rod contact), max |dpos| %.2e |dvel| %.2e" % (checked, in_contact, wp, wv))
    assert checked >= 80 and in_contact >= 10, (checked, in_contact)
    env.close()


def test_batch_properties_and_masks(ctx60):
    """Copies of a context anywhere in a 4096-environment batch evolve bit-identically (other workgroups, other workgroup positions, a ragged last
    workgroup); a masked reset touches only its lanes; bad arguments are refused."""
    from d3il_amd import capi
    n = 4096 + 3
    env = _env(n)
    env.start()
    ctx = ctx60[np.arange(n) % 60]
    env.reset(context=ctx)
    ctx_t = torch.as_tensor(ctx, device=env.device)
    des = env.robot_state().clone()
    for t in range(100):
        des = _script(ctx_t, t, des)
        env.step(_action(des))
    torch.cuda.synchronize()
    st, fl, sc = env.get_state()
    assert np.isfinite(st).all() and not (fl & BAD).any()
    for c in (0, 1, 17, 58):
        cols = st[:, c::120] if c % 2 == 0 else st[:, c::120]      # same context AND same script parity (lane parity = context parity for 120 | stride)
        assert (cols == cols[:, :1]).all(), c
    before = st.copy()
    mask = torch.zeros(n, dtype=torch.uint8, device=env.device)
    mask[5] = 1; mask[4097] = 1
    env.reset(mask=mask, context=ctx)
    torch.cuda.synchronize()
    st2, fl2, sc2 = env.get_state()
    keep = np.ones(n, bool); keep[[5, 4097]] = False
    assert (st2[:, keep] == before[:, keep]).all() and sc2[5] == 0 and sc2[4097] == 0 and sc2[6] == 100
    with pytest.raises(ValueError):
        env.reset(context=ctx[:10])
    with pytest.raises(ValueError):
        env.step(torch.zeros(n, 6, dtype=torch.float64, device=env.device))
    with pytest.raises(capi.D3ilError):
        env.count_metrics()
    env.close()


def test_sim_tables_against_whole_oracle_episodes(ctx60):
    """Aligning_Sim (the C ABI) on the 60 reference contexts with a closed-loop scripted policy, against whole oracle episodes of the same policy:
    identical (success, mode) per context and identical integer metric tables; runs to the step cap or to success."""
    from d3il_amd.agents import ScriptedAlignPolicy
    from d3il_amd.simulation.aligning_sim import Aligning_Sim
    from tests import oracle_episodes as oe
    sim = Aligning_Sim(seed=0, device="cuda:0", render=False, n_cores=1, n_contexts=60, n_trajectories_per_context=1, max_steps_per_episode=400)
    ret = sim.test_agent(ScriptedAlignPolicy(inside=np.arange(60) % 2 == 0, device="cuda:0"))
    r = sim.last_rollout
    # the reference's return signature (aligning_sim.py:205): (success_rate, mode_encoding[n_contexts, n_trajectories])
    assert len(ret) == 2 and isinstance(ret[0], float) and tuple(ret[1].shape) == (60, 1) and ret[0] == r["success_rate"]
    assert not (r["flags"].cpu().numpy() & BAD).any()
    dev_rows = list(zip(r["success"].cpu().numpy().astype(bool).tolist(), r["mode"].cpu().numpy().tolist()))
    env = _env(1)
    q0 = env.start()[0]
    env.close()
    res = oe.run_many(oe.aligning_episode, [(i, ctx60[i], q0, 400, i % 2 == 0) for i in range(60)])
    orc_rows = [(s, m) for _, s, m, _ in res]
    diff = [i for i in range(60) if dev_rows[i] != orc_rows[i]]
    print("aligning: %d of 60 contexts with identical (success, mode); successes device %d / oracle %d; modes %s; differing %s" % (
        60 - len(diff), sum(s for s, _ in dev_rows), sum(s for s, _ in orc_rows), sorted({m for _, m in orc_rows}), [(i, dev_rows[i], orc_rows[i]) for i in diff]))
    assert not diff
    assert {m for _, m in orc_rows} == {0, 1}
