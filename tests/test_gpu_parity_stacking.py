"""Stacking: the HIP path through the C ABI against the CPU oracle (reset on reference contexts, scripted pick-and-place followed
by the oracle, one-step parity from mid-episode device states incl. grasp contacts, batch invariance, auto-reset, Stacking_Sim)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SK_BOX_Z = 28 + 2          # state row of the red box's z
def CubeStackingVecEnv_(n, cap):
    from d3il_amd.envs.stacking import CubeStackingVecEnv
    return CubeStackingVecEnv(n, device=0, max_steps_per_episode=cap)


BAD = (1 << 16) | (1 << 18) | (1 << 19)          # solver fail, contact overflow, off table


@pytest.fixture(scope="module")
def stack_js():
    from d3il_amd.model import blob
    return blob.load_json("stacking")


@pytest.fixture(scope="module")
def stack_blob(stack_js):
    from d3il_amd.model import blob
    return blob.pack(stack_js)


@pytest.fixture(scope="module")
def ctx100():
    return np.load(os.path.join(ROOT, "d3il_amd", "data", "stacking_test_contexts.npy"))


def _env(n, **kw):
    from d3il_amd.envs.stacking import CubeStackingVecEnv
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    return CubeStackingVecEnv(n, device=0, **kw)


def test_reset_matches_oracle_on_reference_contexts(stack_blob, ctx100):
    from oracle.oracle import Oracle
    n = 100
    env = _env(n)
    q0, it, err = env.start()
    assert it == 72
    obs = env.reset(context=ctx100).cpu().numpy()
    st, fl, sc = env.get_state()
    assert env.obs.shape == (n, 12) and env.state_rows == 94 and env.robot_state().shape == (n, 8)
    o = Oracle(stack_blob)
    o.env_start(q0)
    for e in range(0, n, 7):
        oo = o.stack_reset(ctx100[e])
        np.testing.assert_array_equal(obs[e], oo)
        np.testing.assert_allclose(st[:67, e], o.stack_state(), atol=1e-10, rtol=0)
        np.testing.assert_allclose(env.robot_state()[e].cpu().numpy(), o.stack_robot_state(), atol=1e-10)
        assert sc[e] == 0 and not (fl[e] & BAD)
    env.close()


def test_pick_and_place_followed_by_the_oracle(stack_js, stack_blob, ctx100):
    """Every lane runs the scripted pick-and-place of its context's red box; four lanes are followed by the oracle through approach,
    grasp (finger-tip + finger-hull contacts, condim 4), lift, carry and release: all state rows incl. velocities."""
    from d3il_amd.controllers.scripted_stacking import build_trajectory
    from oracle.oracle import Oracle
    ids = [0, 3, 11, 42]
    n = 48
    env = _env(n)
    q0, _, _ = env.start()
    ctx = ctx100[[ids[i % 4] for i in range(n)]]
    env.reset(context=ctx)
    trajs = [build_trajectory(stack_js, q0, ctx100[i], n_boxes=1, speed=0.8) for i in ids]
    T = min(len(t) for t in trajs)
    oracles = []
    for i in ids:
        o = Oracle(stack_blob); o.env_start(q0); o.stack_reset(ctx100[i]); oracles.append(o)
    worst, lifted = 0.0, 0.0
    for t in range(T):
        act = torch.as_tensor(np.stack([trajs[i % 4][t] for i in range(n)]), dtype=torch.float64, device=env.device).contiguous()
        obs, rew, done, info = env.step(act)
        torch.cuda.synchronize()
        st, fl, sc = env.get_state()
        assert not (fl & BAD).any(), "flags at step %d: %s" % (t, hex(int(np.bitwise_or.reduce(fl))))
        for k in range(4):
            oo, do, io = oracles[k].stack_step(trajs[k][t])
            e = k + 4 * (t % 12)                      # a different lane of the same context every step: results do not depend on the lane
            err = float(np.abs(st[:67, e] - oracles[k].stack_state()).max())
            worst = max(worst, err)
            assert err < 1e-5, (t, k, err)
            assert bool(done[e]) == do and bool(info["success"][e]) == io["success"]
            from d3il_amd.envs.stacking import mode_string
            assert mode_string(int(info["mode"][e])) == io["mode"]
            lifted = max(lifted, float(oo[2]))
    assert lifted > 0.08 and worst < 1e-5, (lifted, worst)
    # lanes with the same context are bit-identical
    for k in range(4):
        ref = st[:67, k]
        for e in range(k + 4, n, 4):
            assert np.array_equal(st[:67, e], ref)
    env.close()


def test_full_size_results_do_not_depend_on_lane_or_workgroup(stack_js, ctx100):
    """4096 environments (BASELINE config 5's per-GPU size) through a grasp-and-lift: every lane of a context ends bit-identical, whether
    the four environments of a workgroup belong to four different contexts (cooperative solves of different sizes side by side,
    lane-per-pair collision with mixed jobs) or to one."""
    from d3il_amd.controllers.scripted_stacking import build_trajectory
    ids = [0, 3, 11, 42]
    n = 4096
    finals = []
    for mixed in (True, False):
        which = (np.arange(n) % 4) if mixed else (np.arange(n) // (n // 4))
        env = _env(n)
        q0, _, _ = env.start()
        env.reset(context=ctx100[[ids[w] for w in which]])
        trajs = [build_trajectory(stack_js, q0, ctx100[i], n_boxes=1, speed=0.8) for i in ids]
        T = min(min(len(t) for t in trajs), 90)                 # approach, grasp, lift
        sel = torch.as_tensor(which, device=env.device)
        for t in range(T):
            act = torch.as_tensor(np.stack([tr[t] for tr in trajs]), dtype=torch.float64, device=env.device)[sel].contiguous()
            env.step(act)
        torch.cuda.synchronize()
        st, fl, sc = env.get_state()
        assert not (fl & BAD).any()
        per_ctx = []
        for k in range(4):
            lanes = np.nonzero(which == k)[0]
            ref = st[:67, lanes[0]]
            assert all(np.array_equal(st[:67, e], ref) for e in lanes[1:])
            per_ctx.append(ref.copy())
        finals.append(per_ctx)
        assert max(float(p[SK_BOX_Z]) for p in per_ctx) > 0.05          # the red boxes are off the table
        env.close()
    for k in range(4):
        assert np.array_equal(finals[0][k], finals[1][k])


@pytest.mark.parametrize("n", [1, 5, 4097])
def test_ragged_batches_masked_reset_and_bad_actions(stack_js, ctx100, n):
    """Batch sizes that do not fill the last workgroup (4 environments each), a masked reset of single environments, a NaN / Inf action:
    the affected lanes are handled, every other lane equals the lane of a plain run bit for bit."""
    from d3il_amd import capi
    env, ref = _env(n), _env(n)
    q0, _, _ = env.start(); ref.start()
    ctx = ctx100[np.arange(n) % 7]
    env.reset(context=ctx); ref.reset(context=ctx)
    a0 = torch.cat([torch.as_tensor(q0, dtype=torch.float64, device=env.device).expand(n, 7), torch.ones(n, 1, dtype=torch.float64, device=env.device)], dim=1)
    acts = [(a0 + 0.003 * (t + 1) * torch.tensor([1, -1, 1, -1, 1, -1, 1, 0], dtype=torch.float64, device=env.device)).contiguous() for t in range(6)]
    for t in range(3):
        env.step(acts[t]); ref.step(acts[t])
    torch.cuda.synchronize()
    s_env, _, _ = env.get_state(); s_ref, _, _ = ref.get_state()
    assert np.array_equal(s_env[:67], s_ref[:67])
    # masked reset of the last environment (and of environment 2 when there is one)
    mask = torch.zeros(n, dtype=torch.uint8, device=env.device); mask[n - 1] = 1
    if n > 2:
        mask[2] = 1
    env.reset(mask=mask)
    torch.cuda.synchronize()
    s1, f1, c1 = env.get_state()
    fresh = _env(n); fresh.start(); fresh.reset(context=ctx); torch.cuda.synchronize()
    s0, _, _ = fresh.get_state(); fresh.close()
    hit = mask.cpu().numpy().astype(bool)
    assert np.array_equal(s1[:67, hit], s0[:67, hit]) and (c1[:n][hit] == 0).all()            # restarted on their own context
    assert np.array_equal(s1[:67, ~hit], s_ref[:67, ~hit]) and (c1[:n][~hit] == 3).all()       # the others untouched
    # a NaN / Inf action: flagged and terminated, state finite, neighbours of the same workgroup unaffected
    bad = acts[3].clone(); bad[0, 2] = float("nan")
    if n > 4:
        bad[4, 7] = float("inf")
    env.step(bad)
    good = env.robot_state()          # (forces the kernel to finish before the next reset)
    torch.cuda.synchronize()
    s2, f2, _ = env.get_state()
    assert np.isfinite(s2[:67]).all()
    flagged = (f2[:n] & capi.FLAG_SOLVER_FAIL) != 0
    assert flagged[0] and flagged.sum() == (2 if n > 4 else 1) and (f2[:n][flagged] & capi.FLAG_TERMINATED).all()
    if n > 1:      # lane 1 shares the workgroup of lane 0: compare with a run that has no bad action
        ref.reset(mask=mask); ref.step(acts[3]); torch.cuda.synchronize()
        s3, _, _ = ref.get_state()
        ok = ~flagged
        assert np.array_equal(s2[:67, ok], s3[:67, ok])
    env.close(); ref.close()


def test_state_snapshot_restores_the_rollout_exactly(stack_js, ctx100):
    """d3il_get_state / d3il_set_state carry everything a rollout depends on (arm, boxes, the solver's warm start, flags, counters):
    restoring a mid-grasp snapshot and repeating the steps reproduces the continuation bit for bit."""
    from d3il_amd.controllers.scripted_stacking import build_trajectory
    n = 9
    env = _env(n)
    q0, _, _ = env.start()
    env.reset(context=ctx100[np.arange(n) % 3])
    trajs = [build_trajectory(stack_js, q0, ctx100[i], n_boxes=1, speed=0.8) for i in range(3)]

    def act(t):
        return torch.as_tensor(np.stack([trajs[i % 3][t] for i in range(n)]), dtype=torch.float64, device=env.device).contiguous()

    for t in range(70):           # into the grasp
        env.step(act(t))
    torch.cuda.synchronize()
    snap = env.get_state()
    for t in range(70, 80):
        env.step(act(t))
    torch.cuda.synchronize()
    a_state, a_flags, a_steps = env.get_state()
    env.set_state(*snap)
    for t in range(70, 80):
        env.step(act(t))
    torch.cuda.synchronize()
    b_state, b_flags, b_steps = env.get_state()
    assert np.array_equal(a_state, b_state) and np.array_equal(a_flags, b_flags) and np.array_equal(a_steps, b_steps)
    assert not (a_flags & BAD).any()
    env.close()


@pytest.mark.parametrize("strict", [0, 1])
def test_one_step_parity_from_mid_episode_states(stack_js, stack_blob, ctx100, strict):
    from d3il_amd.controllers.scripted_stacking import build_trajectory
    from oracle.oracle import Oracle
    ids = [1, 5, 17, 60, 77, 93]
    n = len(ids)
    env = _env(n)
    env.set_option("solver_strict", strict)
    q0, _, _ = env.start()
    env.reset(context=ctx100[ids])
    trajs = [build_trajectory(stack_js, q0, ctx100[i], n_boxes=2, speed=1.0) for i in ids]
    T = min(len(t) for t in trajs)
    o = Oracle(stack_blob); o.env_start(q0); o.stack_reset(ctx100[0])
    worst_p = worst_v = 0.0
    vel = list(range(9, 18)) + [28 + 13 * b + 7 + k for b in range(3) for k in range(6)]
    pos = [r for r in range(67) if r not in vel and not (18 <= r < 25)]
    checked = grasp = 0
    for t in range(T):
        act = torch.as_tensor(np.stack([trajs[k][t] for k in range(n)]), dtype=torch.float64, device=env.device).contiguous()
        if t % 6 == 5:
            torch.cuda.synchronize()
            st0, fl0, sc0 = env.get_state()
        env.step(act)
        if t % 6 == 5:
            torch.cuda.synchronize()
            st1, fl1, sc1 = env.get_state()
            for k in (t // 6 % n, (t // 6 + 3) % n):
                nm = int(fl0[k] & 3)
                o.stack_set_state(st0[:67, k], step=int(sc0[k]), terminated=bool(fl0[k] & (1 << 12)), min_inds=[int((fl0[k] >> (2 + 2 * i)) & 3) for i in range(nm)])
                o.stack_step(trajs[k][t])
                so = o.stack_state()
                worst_p = max(worst_p, float(np.abs(st1[pos, k] - so[pos]).max())); worst_v = max(worst_v, float(np.abs(st1[vel, k] - so[vel]).max()))
                grasp += sum(1 for c in o.contacts() if c[9] > 40) >= 4
                checked += 1
    assert checked > 40 and grasp > 5
    assert worst_p < 1e-7 and worst_v < 1e-5, (worst_p, worst_v)       # north star 1e-4
    env.close()


def test_empty_gripper_closes_on_itself_like_the_oracle(stack_blob, ctx100):
    """close_fingers with nothing between the fingers: the finger <-> finger pairs (tip boxes: box-box, finger hulls: three MPR jobs on
    eight-lane groups) carry the closing force - the collision group of the device kernel no other test reaches."""
    from oracle.oracle import Oracle
    n = 6
    env = _env(n)
    q0, _, _ = env.start()
    env.reset(context=ctx100[np.arange(n) % 2])
    oracles = []
    for k in range(2):
        o = Oracle(stack_blob); o.env_start(q0); o.stack_reset(ctx100[k]); oracles.append(o)
    a = np.concatenate([q0, [0.0]])
    act = torch.as_tensor(np.tile(a, (n, 1)), dtype=torch.float64, device=env.device).contiguous()
    worst = 0.0
    for t in range(14):
        env.step(act)
        torch.cuda.synchronize()
        st, fl, sc = env.get_state()
        assert not (fl & BAD).any()
        for k in range(2):
            oracles[k].stack_step(a)
            worst = max(worst, float(np.abs(st[:67, k + 2 * (t % 3)] - oracles[k].stack_state()).max()))
    assert worst < 1e-6, worst
    w = float(env.robot_state()[0, 7])
    assert 0.0 < w < 0.002, w                     # the contact equilibrium of the pads, not the joint stops
    env.close()


def test_random_policy_raises_no_solver_failure(ctx100):
    """BASELINE config 5's closed loop with the randomly initialised BESO policy (bench.py --task stacking --policy beso): 4096 environments,
    60 policy steps of small random joint motions with the gripper opening and closing at random.  Round 3 found SOLVER_FAIL flags here
    (about one environment in 1e4 per step, always with a nearly closed empty gripper: the finger <-> finger MPR jobs) that none of the
    scripted tests reached; cause and fix: stack_step.h sk_support1_group_pre / DESIGN section 17.3.  No flag may be raised."""
    import bench
    n = 4096
    env = _env(n)
    env.start()
    env.reset(context=ctx100[np.arange(n) % 16])
    pol = bench._random_beso(env.device)
    last_cmd = env.robot_state().to(torch.float32).clone()
    for t in range(60):
        obs20 = torch.cat((last_cmd, env.obs), dim=1)
        out = pol.predict_batch(obs20).to(torch.float32)
        last_cmd = torch.cat((out[:, :7] + obs20[:, :7], out[:, 7:8]), dim=1)
        env.step(last_cmd.to(torch.float64).contiguous())
    torch.cuda.synchronize()
    st, fl, sc = env.get_state()
    assert np.isfinite(st).all()
    bad = np.nonzero(fl & BAD)[0]
    assert bad.size == 0, "flags raised in environments %s (workgroup positions %s): %s" % (bad[:8].tolist(), (bad[:8] % 4).tolist(), [hex(int(x)) for x in fl[bad[:8]]])
    assert float((st[7] + st[8]).min()) < 0.004, "the run has to reach the closed-gripper regime (finger <-> finger jobs)"
    env.close()


def test_copies_of_an_environment_in_one_workgroup_stay_bit_identical(ctx100):
    """Four copies of the same environment share a workgroup (the four positions are served by different lanes / lane groups of the
    cooperative kernel: collision groups, MPR lane groups, the two solver halves) and receive the same actions - random joint motions while
    the gripper closes on nothing and opens again.  Their states must agree bit for bit at every step, for every workgroup."""
    n, wgs = 1024, 256
    env = _env(n)
    q0, _, _ = env.start()
    wg = np.arange(n) // 4
    env.reset(context=ctx100[wg % 16])
    rng = np.random.default_rng(5)
    cmd = np.tile(np.asarray(q0, dtype=np.float64), (wgs, 1))
    for t in range(70):
        cmd = cmd + rng.uniform(-0.01, 0.01, size=cmd.shape)
        grip = np.where((t + np.arange(wgs)) % 35 < 25, 0.0, 0.08)        # closed for 25 steps, open for 10, phase per workgroup
        act = np.concatenate([cmd, grip[:, None]], axis=1)[wg]
        env.step(torch.as_tensor(act, dtype=torch.float64, device=env.device).contiguous())
        torch.cuda.synchronize()
        st, fl, sc = env.get_state()
        assert not (fl & BAD).any(), (t, np.nonzero(fl & BAD)[0][:8].tolist())
        s4 = st.reshape(st.shape[0], wgs, 4)
        diff = np.nonzero((s4 != s4[:, :, :1]).any(axis=(0, 2)))[0]
        assert diff.size == 0, "step %d: workgroups %s hold copies that differ (positions %s)" % (
            t, diff[:6].tolist(), [np.nonzero((s4[:, w, :] != s4[:, w, :1]).any(axis=0))[0].tolist() for w in diff[:6]])
    assert float((st[7] + st[8]).min()) < 0.004
    env.close()


def test_permuted_batch_gives_every_environment_the_same_trajectory(ctx100):
    """Every environment has its own context and its own open-loop actions (random joint motions, the gripper closing on nothing and opening again);
    the batch runs twice, the second time permuted (other workgroup mates, other workgroup positions, other MPR lane groups, other solver halves).
    Each environment's state must be bit-identical in both runs at every step (tools/gpu_stack_perm.py is the long version of this test)."""
    n, steps = 2048, 150
    rng = np.random.default_rng(5)
    cid = rng.integers(0, 100, size=n)
    phase, period = rng.integers(0, 40, size=n), rng.integers(20, 60, size=n)
    delta = rng.uniform(-0.01, 0.01, size=(steps, n, 7))
    perm = rng.permutation(n)
    runs = []
    for order in (np.arange(n), perm):
        env = _env(n)
        q0, _, _ = env.start()
        env.reset(context=ctx100[cid[order]])
        cmd = np.tile(np.asarray(q0, dtype=np.float64), (n, 1))
        inv = np.argsort(order)
        out = []
        for t in range(steps):
            cmd = cmd + delta[t]
            grip = np.where((t + phase) % period < 0.7 * period, 0.0, 0.08)
            act = np.concatenate([cmd, grip[:, None]], axis=1)[order]
            env.step(torch.as_tensor(act, dtype=torch.float64, device=env.device).contiguous())
            torch.cuda.synchronize()
            st, fl, _ = env.get_state()
            assert not (fl & (1 << 16)).any(), (t, np.nonzero(fl & (1 << 16))[0][:8].tolist())
            out.append(st[:, inv].copy())
        runs.append(out)
        env.close()
    for t in range(steps):
        d = np.nonzero((runs[0][t] != runs[1][t]).any(axis=0))[0]
        assert d.size == 0, "step %d: environments %s differ between the two arrangements (positions %s / %s)" % (t, d[:6].tolist(), (d[:6] % 4).tolist(), (np.argsort(perm)[d[:6]] % 4).tolist())
    assert float((runs[0][-1][7] + runs[0][-1][8]).min()) < 0.004


def test_contact_overflow_is_flagged_and_contained(ctx100):
    """More contacts than the record area holds (three boxes pushed into each other on the table: 3 x 8 box-box + 3 x 4 table contacts):
    the lane raises CON_OVERFLOW, stays finite, and the other lanes of its workgroup are not disturbed."""
    n = 8
    env, ref = _env(n), _env(n)
    q0, _, _ = env.start(); ref.start()
    ctx = ctx100[np.arange(n) % 4]
    env.reset(context=ctx); ref.reset(context=ctx)
    torch.cuda.synchronize()
    st, fl, sc = env.get_state()
    lane = 1
    for b in range(3):            # the three boxes of one lane nearly on top of each other (1 mm offsets), resting height
        o = 28 + 13 * b
        st[o:o + 3, lane] = [0.5 + 0.001 * b, 0.1 - 0.001 * b, st[28 + 2, lane]]
        st[o + 3:o + 7, lane] = [1, 0, 0, 0]
        st[o + 7:o + 13, lane] = 0
    env.set_state(st, fl, sc)
    a = torch.cat([torch.as_tensor(q0, dtype=torch.float64, device=env.device).expand(n, 7), torch.ones(n, 1, dtype=torch.float64, device=env.device)], dim=1).contiguous()
    env.step(a); ref.step(a)
    torch.cuda.synchronize()
    s1, f1, _ = env.get_state(); s0, f0, _ = ref.get_state()
    assert f1[lane] & (1 << 18), hex(int(f1[lane]))
    assert np.isfinite(s1[:67]).all()
    others = np.arange(n) != lane
    assert np.array_equal(s1[:67, others], s0[:67, others]) and np.array_equal(f1[:n][others], f0[:n][others])
    env.close(); ref.close()


def test_device_auto_reset_restarts_lanes_on_their_contexts(ctx100):
    """d3il_auto_reset for Stacking: finished lanes (episode cap) are tallied per context, restarted on the context they were created
    with, marked in last_reset; the restarted state equals a fresh reset of that context bit for bit."""
    n, cap = 14, 6
    env = CubeStackingVecEnv_(n, cap)
    q0, _, _ = env.start()
    ctx_id = np.arange(n) % 3
    env.reset(context=ctx100[ctx_id])
    table = env.set_tally(3, torch.as_tensor(ctx_id, dtype=torch.int32))
    counts = torch.zeros(2, dtype=torch.int64, device=env.device)
    torch.cuda.synchronize()
    fresh, _, _ = env.get_state()
    act = torch.cat([torch.as_tensor(q0, dtype=torch.float64, device=env.device).expand(n, 7) + 0.01, torch.ones(n, 1, dtype=torch.float64, device=env.device)], dim=1).contiguous()
    resets = 0
    for t in range(20):
        env.step(act)
        env.auto_reset(counts)
        torch.cuda.synchronize()
        lr = env.last_reset.cpu().numpy().astype(bool)
        if lr.any():
            assert lr.all()                       # all lanes share the cap, so they finish together
            st, fl, sc = env.get_state()
            assert np.array_equal(st[:67], fresh[:67]) and (sc[:n] == 0).all()
            resets += 1
    assert resets == 3                            # done is raised at step index cap - 1, i.e. by the step() calls 6, 12, 18
    tb = table.cpu().numpy()
    assert tb[:, 0].tolist() == [3 * int((ctx_id == c).sum()) for c in range(3)] and tb[:, 1].sum() == 0
    assert counts.cpu().tolist() == [3 * n, 0]
    env.close()


def test_auto_reset_tally_and_sim(stack_js, ctx100):
    from d3il_amd.agents import ScriptedStackPolicy
    from d3il_amd.controllers.scripted_stacking import build_trajectory
    from d3il_amd.simulation.stacking_sim import Stacking_Sim
    # Stacking_Sim with the scripted policy on 4 contexts x 2 rollouts: the pick-and-place succeeds (slow script) and the metric tail runs
    sim = Stacking_Sim(seed=0, device="cuda:0", render=False, n_contexts=4, n_trajectories_per_context=2, max_steps_per_episode=900)
    from d3il_amd.envs.stacking import CubeStackingVecEnv
    probe = CubeStackingVecEnv(1, device=0)
    q0, _, _ = probe.start(); probe.close()
    tables = [build_trajectory(stack_js, q0, sim.test_contexts[c], speed=0.5) for c in range(4)]
    pol = ScriptedStackPolicy(tables, np.arange(8) // 2, device="cuda:0")
    succ, modes = sim.test_agent(pol)
    assert succ.shape == (4, 2) and modes.shape == (4, 2)
    r = sim.last_rollout
    assert not (r["flags"].cpu().numpy() & BAD).any()
    assert float(succ.mean()) >= 0.75 and r["metrics"]["successes_1_box"] == 1.0
    assert torch.equal(succ[:, 0], succ[:, 1]) and torch.equal(modes[:, 0], modes[:, 1])     # identical rollouts of one context agree bit for bit


def test_raw_c_abi_stacking_create_reset_step(stack_blob, ctx100):
    """The Stacking row of the boundary through RAW ctypes (no Python env class): d3il_create(task 3) -> start -> reset(contexts [n][21]) ->
    step(actions [n][8]) -> get_buffers / get_state with the shapes and constants include/d3il_rollout.h documents, checked against the oracle."""
    import ctypes as C
    from d3il_amd import capi
    from oracle.oracle import Oracle
    L = capi.load()
    n = 6
    h = C.c_void_p()
    assert L.d3il_create(capi.TASK_STACKING, n, 0, C.byref(stack_blob), C.sizeof(stack_blob), C.byref(h)) == 0, L.d3il_last_error()
    b = capi.Buffers()
    assert L.d3il_get_buffers(h, C.byref(b)) == 0
    assert (b.n_envs, b.stride, b.obs_dim, b.action_dim, b.state_rows, b.n_info_f64) == (n, 64, 12, 8, capi.STACK_STATE_F64, 1)
    q0 = np.load(os.path.join(ROOT, "tests", "golden", "ref_offline_ik.npz"))["stacking__traj_last"].copy()
    ctx = torch.as_tensor(ctx100[:n], dtype=torch.float64, device="cuda:0").contiguous()
    assert L.d3il_reset(h, None, C.c_void_p(ctx.data_ptr()), None) == -6                       # env.start() first
    assert L.d3il_start(h, q0.ctypes.data_as(C.c_void_p)) == 0
    assert L.d3il_reset(h, None, None, None) == -1 and b"contexts" in L.d3il_last_error()      # Stacking needs contexts
    assert L.d3il_reset(h, None, C.c_void_p(ctx.data_ptr()), None) == 0
    act7 = torch.zeros(n, 7, dtype=torch.float64, device="cuda:0")
    assert L.d3il_policy_action(h, 42, 0, 0, C.c_void_p(act7.data_ptr()), None) == -5           # 7-wide random policy: not for the 8-wide Stacking action
    a = np.concatenate([q0 + 0.01, [1.0]])
    act = torch.as_tensor(np.tile(a, (n, 1)), dtype=torch.float64, device="cuda:0").contiguous()
    o = Oracle(stack_blob); o.env_start(q0); o.stack_reset(ctx100[2])
    for t in range(3):
        assert L.d3il_step(h, C.c_void_p(act.data_ptr()), None) == 0
        oo, od, oi = o.stack_step(a)
    torch.cuda.synchronize()
    st = np.zeros((capi.STACK_STATE_F64, n)); fl = np.zeros(n, dtype=np.uint32); sc = np.zeros(n, dtype=np.int32)
    assert L.d3il_get_state(h, st.ctypes.data_as(C.c_void_p), st.shape[0], fl.ctypes.data_as(C.c_void_p), sc.ctypes.data_as(C.c_void_p)) == 0
    assert (sc == 3).all() and not (fl & (capi.FLAG_SOLVER_FAIL | capi.PFLAG_CON_OVERFLOW | capi.PFLAG_OFF_TABLE | capi.SFLAG_HAND_NEAR)).any()
    np.testing.assert_allclose(st[:capi.STACK_STATE_WARM, 2], o.stack_state(), atol=1e-9, rtol=0)
    hobs = np.zeros((n, 12), dtype=np.float32)
    hip = C.CDLL("libamdhip64.so")
    assert hip.hipMemcpy(hobs.ctypes.data_as(C.c_void_p), C.c_void_p(b.obs), C.c_size_t(hobs.nbytes), 2) == 0       # hipMemcpyDeviceToHost
    np.testing.assert_array_equal(hobs[2], oo)
    # box rows are where the header says: red box position = context (x, y), resting height
    assert abs(st[capi.STACK_STATE_BOX + 0, 2] - ctx100[2][0]) < 1e-3 and abs(st[capi.STACK_STATE_BOX + 1, 2] - ctx100[2][1]) < 1e-3
    assert L.d3il_destroy(h) == 0


def test_library_rccl_reduction_single_rank():
    """d3il_comm_* / d3il_reduce_metrics: the library's own RCCL path (one rank is all a one-GPU box can form; RCCL refuses two ranks on one
    GPU): unique id, communicator, in-place int64 all-reduce on the caller's stream, destroy; and the torch-independent error paths."""
    import ctypes as C
    from d3il_amd import capi, distributed as D
    from d3il_amd.envs.stacking import CubeStackingVecEnv
    L = capi.load()
    assert L.d3il_comm_unique_id(None) == -1 and L.d3il_comm_destroy(None) == -1
    env = CubeStackingVecEnv(8, device=0)
    comm = D.LibraryComm(torch.device("cuda:0"))
    assert comm.world == 1 and comm.ranks() == 1 and L.d3il_rccl_available() == 1      # ncclCommCount: what RCCL itself sees
    table = torch.arange(3 * capi.TALLY_ROW, dtype=torch.int64, device="cuda:0")
    ref = table.clone()
    D.reduce_counts(table, comm, env.h)
    torch.cuda.synchronize()
    assert torch.equal(table, ref)                                     # sum over one rank
    D.reduce_counts(table, comm, None)                                 # a rank without environments: no handle, explicit table
    torch.cuda.synchronize()
    assert torch.equal(table, ref) and "1 ranks" in D.LAST_REDUCTION
    assert L.d3il_reduce_metrics(None, comm.comm, None, 0, None) == -1 and L.d3il_comm_count(None, None) == -1
    assert L.d3il_reduce_metrics(env.h, comm.comm, None, 0, None) == -1     # no table registered with d3il_set_tally
    comm.close()
    env.close()


def test_config5_total_size_32768_environments(stack_js, ctx100):
    """BASELINE config 5's TOTAL size (32768 environments; per GPU it is 4096) on one GPU: the engine is not tied to the per-GPU shard size.
    Eight reference contexts tiled, scripted approach - grasp - lift: every state finite, no flag, all 4096 lanes of a context bit-identical and
    equal to the same context in a 64-environment batch."""
    from d3il_amd.controllers.scripted_stacking import build_trajectory
    ids = [0, 3, 11, 42, 57, 64, 80, 99]
    finals = {}
    for n in (64, 32768):
        env = _env(n)
        q0, _, _ = env.start()
        which = np.arange(n) % len(ids)
        env.reset(context=ctx100[[ids[w] for w in which]])
        trajs = [build_trajectory(stack_js, q0, ctx100[i], n_boxes=1, speed=0.8) for i in ids]
        T = min(min(len(t) for t in trajs), 75)
        tab = torch.as_tensor(np.stack([tr[:T] for tr in trajs]), dtype=torch.float64, device=env.device)      # [ctx, T, 8]
        sel = torch.as_tensor(which, device=env.device)
        for t in range(T):
            env.step(tab[sel, t].contiguous())
        torch.cuda.synchronize()
        st, fl, sc = env.get_state()
        assert np.isfinite(st).all() and (sc == T).all()
        assert not (fl & BAD).any(), hex(int(np.bitwise_or.reduce(fl)))
        per_ctx = []
        for k in range(len(ids)):
            lanes = np.nonzero(which == k)[0]
            ref = st[:, lanes[0]]
            assert (st[:, lanes] == ref[:, None]).all()
            per_ctx.append(ref.copy())
        finals[n] = per_ctx
        assert max(float(p[SK_BOX_Z]) for p in per_ctx) > 0.02          # boxes have left the table (rest height 0.011)
        env.close()
    for k in range(len(ids)):
        assert np.array_equal(finals[64][k], finals[32768][k])


def test_palm_pressed_on_a_box_follows_the_oracle(stack_js, stack_blob, ctx100):
    """The box <-> hand-hull pair on the device (groups 13 .. 15 of the lane-per-pair collision: bounding-box cull, MPR by the whole wave over the 773
    hull vertices): the open gripper is lowered until the palm presses on the red box; two contexts followed by the oracle on all state rows;
    SKF_HAND_NEAR is never raised any more (the pair is evaluated, not flagged)."""
    from d3il_amd.controllers.scripted_stacking import build_palm_press
    from oracle.oracle import Oracle
    names = [g["name"] for g in stack_js["geoms"]]
    hand_g = names.index("panda_rb0_hand:geom2")
    ids = [0, 42]
    n = 8
    env = _env(n)
    q0, _, _ = env.start()
    env.reset(context=ctx100[[ids[i % 2] for i in range(n)]])
    trajs = [build_palm_press(stack_js, q0, ctx100[i]) for i in ids]
    oracles = []
    for i in ids:
        o = Oracle(stack_blob); o.env_start(q0); o.stack_reset(ctx100[i]); oracles.append(o)
    worst, n_hand = 0.0, 0
    for t in range(len(trajs[0])):
        act = torch.as_tensor(np.stack([trajs[i % 2][t] for i in range(n)]), dtype=torch.float64, device=env.device).contiguous()
        env.step(act)
        torch.cuda.synchronize()
        st, fl, sc = env.get_state()
        assert not (fl & (BAD | (1 << 20))).any(), hex(int(np.bitwise_or.reduce(fl)))
        for k in range(2):
            oracles[k].stack_step(trajs[k][t])
            e = k + 2 * (t % 4)
            err = float(np.abs(st[:67, e] - oracles[k].stack_state()).max())
            worst = max(worst, err)
            assert err < 1e-5, (t, k, err)
            if any(int(c[8]) == hand_g or int(c[9]) == hand_g for c in oracles[k].contacts()):
                n_hand += 1
    assert n_hand >= 20 and worst < 1e-5, (n_hand, worst)
    env.close()
