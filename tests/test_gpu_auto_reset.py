"""Device auto-reset with per-context episode tally (Pushing / Sorting), the guards added in round 2 (NaN actions, one model per
device, empty handles) - through the C ABI on the GPU."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = np.load(os.path.join(ROOT, "tests", "golden", "ref_offline_ik.npz"))


def _action(des, z):
    quat = torch.tensor([0.0, 1, 0, 0], dtype=torch.float64, device=des.device).expand(des.shape[0], 4)
    return torch.cat([des, z, quat], dim=1).contiguous()


def test_pushing_auto_reset_restarts_finished_lanes_from_their_context():
    from d3il_amd import capi
    from d3il_amd.envs.pushing import BlockPushVecEnv
    ctx60 = np.load(os.path.join(ROOT, "d3il_amd", "data", "pushing_test_contexts.npy"))
    n = 90
    ids = np.arange(n) % 7
    env = BlockPushVecEnv(n, device=0, max_steps_per_episode=6)
    env.set_init_qpos(G["avoiding__traj_last"].copy())
    env.reset(context=ctx60[ids])
    torch.cuda.synchronize()
    st_reset, fl_reset, _ = env.get_state()
    table = env.set_tally(7, torch.as_tensor(ids, dtype=torch.int32))
    # stagger: lane i is i % 6 steps into its episode
    env.step_count[:n] = torch.as_tensor(np.arange(n) % 6, dtype=torch.int32, device=env.device)
    env.policy_begin()
    counts = torch.zeros(2, dtype=torch.int64, device=env.device)
    finished_total = 0
    for t in range(9):
        des = env.policy_des[:2, :n].t().clone() + 0.001
        env.policy_des[:2, :n] = des.t()
        env.step(_action(des, env.policy_des[2:3, :n].t().clone()))
        done = env.done.clone().cpu().numpy().astype(bool)
        env.auto_reset(counts)
        torch.cuda.synchronize()
        finished_total += int(done.sum())
        st, fl, sc = env.get_state()
        np.testing.assert_array_equal(env.last_reset.cpu().numpy().astype(bool), done)
        assert np.all(sc[done] == 0) and not env.done.any()
        # a lane that was reset is bit-identical to the state d3il_reset produced for its context
        np.testing.assert_array_equal(st[:68][:, done], st_reset[:68][:, done])
        # and its harness set-point is re-latched to the TCP
        np.testing.assert_array_equal(env.policy_des[:3, :n].cpu().numpy()[:, done], st[25:28][:, done])
    tb = table.cpu().numpy()
    assert int(counts[0]) == finished_total == int(tb[:, 0].sum()) and finished_total >= n
    # every context row has counted its own lanes only
    lanes_per_ctx = np.bincount(ids, minlength=7)
    assert np.all(tb[:, 0] >= lanes_per_ctx) and tb[:, 1].sum() == int(counts[1]) == 0
    assert not (fl & (capi.FLAG_SOLVER_FAIL | capi.PFLAG_CON_OVERFLOW)).any()
    env.close()


def test_sorting_auto_reset_and_tally_codes():
    from d3il_amd.envs.sorting import SortingVecEnv, sample_contexts
    n = 40
    ctx = sample_contexts(n, 4, seed=2)
    env = SortingVecEnv(n, device=0, max_steps_per_episode=4)
    env.set_init_qpos(G["sorting__traj_last"].copy())
    env.reset(context=ctx)
    torch.cuda.synchronize()
    st_reset, _, _ = env.get_state()
    table = env.set_tally(1)
    env.policy_begin()
    z = env.robot_state()[:, 2:3].clone()
    des = env.obs[:, :2].to(torch.float64).clone()
    for t in range(4):
        env.step(_action(des, z))
    assert bool(env.done.all())
    env.auto_reset()
    torch.cuda.synchronize()
    st, fl, sc = env.get_state()
    np.testing.assert_array_equal(st[:94], st_reset[:94])
    assert np.all(sc == 0) and (env.mode.cpu().numpy() == 240).all()
    tb = table.cpu().numpy()
    assert tb[0, 0] == n and tb[0, 1] == 0 and tb[0, 2:].sum() == 0
    env.close()


def test_nan_action_is_flagged_not_propagated():
    from d3il_amd import capi
    from d3il_amd.envs.avoiding import ObstacleAvoidanceVecEnv
    from d3il_amd.envs.pushing import BlockPushVecEnv
    ctx60 = np.load(os.path.join(ROOT, "d3il_amd", "data", "pushing_test_contexts.npy"))
    from d3il_amd.envs.sorting import SortingVecEnv, sample_contexts
    for cls in (ObstacleAvoidanceVecEnv, BlockPushVecEnv, SortingVecEnv):
        n = 70
        env = cls(n, device=0)
        env.set_init_qpos(G["sorting__traj_last" if cls is SortingVecEnv else "avoiding__traj_last"].copy())
        if cls is BlockPushVecEnv:
            env.reset(context=ctx60[np.arange(n) % 60])
        elif cls is SortingVecEnv:
            env.reset(context=sample_contexts(n, 4, seed=3))
        else:
            env.reset()
        rs = env.robot_state().clone()
        act = _action(rs[:, :2].clone(), rs[:, 2:3].clone())
        act[3, 1] = float("nan"); act[64, 0] = float("inf")
        env.step(act)
        torch.cuda.synchronize()
        st, fl, sc = env.get_state()
        assert np.isfinite(st[:42]).all()
        bad = (fl & capi.FLAG_SOLVER_FAIL) != 0
        assert bad[3] and bad[64] and bad.sum() == 2
        assert (fl[[3, 64]] & capi.FLAG_TERMINATED).all()
        env.step(_action(rs[:, :2].clone(), rs[:, 2:3].clone()))
        torch.cuda.synchronize()
        assert env.done.cpu().numpy()[[3, 64]].all() and env.done.sum() == 2
        env.close()


def test_handles_of_different_generic_models_live_side_by_side():
    """Sorting-4, Sorting-2 and Pushing all run on the generic engine, whose constants sit in ONE __constant__ object per device.  Until round 4 a second
    model was refused while a handle of another lived; since round 5 every launch checks which model the device holds and reloads it (rollout.hip GenLaunch).
    Handles of three models stepped in turn give the states of the same handles stepped alone."""
    from d3il_amd.envs.pushing import BlockPushVecEnv, sample_contexts as push_contexts
    from d3il_amd.envs.sorting import SortingVecEnv, sample_contexts

    def make():
        a = SortingVecEnv(24, device=0, num_boxes=4); a.start(); a.reset(context=sample_contexts(24, 4, seed=5))
        b = SortingVecEnv(24, device=0, num_boxes=2); b.start(); b.reset(context=sample_contexts(24, 2, seed=6))
        c = BlockPushVecEnv(24, device=0); c.start(); c.reset(context=push_contexts(24, seed=7))
        return [a, b, c]

    def act(env, t):
        rs = env.robot_state()
        return _action(rs[:, :2] + torch.tensor([0.0, 0.004 * (t + 1)], dtype=torch.float64, device=rs.device), rs[:, 2:3].clone())

    envs = make()
    acts = [[act(e, t) for t in range(4)] for e in envs]
    for t in range(4):
        for e, a in zip(envs, acts):      # interleaved: every launch follows one of another model
            e.step(a[t])
    torch.cuda.synchronize()
    mixed = [e.get_state()[0].copy() for e in envs]
    for e in envs:
        e.close()
    envs = make()
    for e, a in zip(envs, acts):          # one after the other
        for t in range(4):
            e.step(a[t])
    torch.cuda.synchronize()
    for m, e in zip(mixed, envs):
        assert np.array_equal(m, e.get_state()[0])
        e.close()


def test_captured_rollout_step_equals_the_eight_calls():
    """Option graph_rollout: d3il_random_rollout_step as ONE HIP graph launch (policy kernel with the step counter in device memory, step kernel between an
    event pair, mask copy, tally, auto-reset, counter + 1) gives bit-identical states, flags, tallies and episode counts to the uncaptured sequence - through
    auto-resets, a jump of the caller's step counter, a timing switch (re-capture) and with the launch durations accounted for."""
    from d3il_amd.envs.avoiding import ObstacleAvoidanceVecEnv
    n = 192
    out = []
    for graph in (0, 1, 2):      # 0: the separate launches, 1: captured graph, 2: fused tail (option fuse_rollout_tail: step kernel + one kernel for everything between two steps)
        stream = torch.cuda.Stream(0)
        with torch.cuda.stream(stream):
            env = ObstacleAvoidanceVecEnv(n, device=0, max_steps_per_episode=40)
            env.bind_stream(stream)
            env.start(); env.reset(); env.policy_begin()
            table = env.set_tally(1, None)
            episodes = torch.zeros(2, dtype=torch.int64, device=env.device)
            actions = torch.zeros(n, 7, dtype=torch.float64, device=env.device)
            if graph == 1:
                env.set_option("graph_rollout", 1)
            if graph == 2:
                env.set_option("fuse_rollout_tail", 1)
            t = 0
            for k in range(70):
                if k == 30:
                    t += 5                      # the caller skips counter values: the device copy follows
                if k == 45:
                    env.set_timing(True)        # changes what a step launches: graphs are dropped and captured again
                    env.random_rollout_prepare(7, 1000, t, actions, episodes)
                env.random_rollout_step(7, 1000, t, actions, episodes)
                t += 1
            stream.synchronize()
            tsum, tmin, tmax, cnt = env.timing_stats()
            assert (cnt == 25 if graph != 1 else 2 <= cnt <= 4) and 0 < tmin <= tmax      # captured form: every eighth launch goes through the event ring
            st, fl, sc = env.get_state()
            out.append((st.copy(), fl.copy(), sc.copy(), table.cpu().numpy().copy(), episodes.cpu().numpy().copy(), actions.cpu().numpy().copy()))
            env.close()
    for mode, other in enumerate(out[1:], 1):
        for k, (a, b) in enumerate(zip(out[0], other)):
            if mode == 2 and k == 5:
                continue                  # fused tail: `actions` already holds the NEXT step's action
            assert np.array_equal(a, b), (mode, k)
    assert out[0][4][0] >= n          # every environment finished at least one 40-step episode
