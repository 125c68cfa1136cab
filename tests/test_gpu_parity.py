"""Parity tests proper: the HIP path, called through the C ABI, against the CPU oracle and the committed
golden fixtures.  Tolerance: the north star asks for 1e-4 on trajectory state; oracle and kernels are both
f64 and agree to ~1e-10, the tests assert 1e-8.  Integer outputs (done, success, mode, counters) bit-exact."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-8


@pytest.fixture(scope="module")
def lib_loaded():
    from d3il_amd import capi
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    return capi.load()      # raises loudly if the HIP extension is missing


def _env(n, **kw):
    from d3il_amd.envs.avoiding import ObstacleAvoidanceVecEnv
    return ObstacleAvoidanceVecEnv(n, device=0, **kw)


def test_start_computes_reference_init_qpos(lib_loaded):
    env = _env(64)
    q, iters, err = env.start()
    g = np.load(os.path.join(G, "ref_offline_ik.npz"))
    assert iters == int(g["avoiding__iters"])
    np.testing.assert_allclose(q, g["avoiding__traj_last"], atol=1e-12)
    env.close()


@pytest.mark.parametrize("fast", [1, 0, 2])
@pytest.mark.parametrize("name", ["random", "collide", "succeed", "zigzag"])
def test_golden_rollouts(lib_loaded, name, fast):
    """Replay the committed oracle rollouts: every env of the batch gets the same actions.  fast = 1: the default kernel (three waves: controller,
    physics, rare constraint paths - `collide` runs the rod contact through the serving wave), 2: the two-wave form used above 256 workgroups
    (rare paths inlined in the physics wave), 0: the fused one-wave kernel with the Jacobi controller path."""
    g = np.load(os.path.join(G, "oracle_avoiding_rollout.npz"))
    n = 128
    env = _env(n)
    env.set_option("ik_fast_path", 1 if fast else 0)
    if fast == 2:
        env.set_option("serve_wave_max_workgroups", 0)
    env.set_init_qpos(g["init_qpos"])
    obs = env.reset()
    torch.cuda.synchronize()
    st, fl, sc = env.get_state()
    np.testing.assert_allclose(st[:, 0], g[name + "__states"][0], atol=TOL)
    assert np.array_equal(obs[0].cpu().numpy(), g[name + "__obs"][0])
    acts = g[name + "__actions"]
    for t in range(len(acts)):
        a = torch.as_tensor(np.tile(acts[t], (n, 1)), dtype=torch.float64, device=env.device).contiguous()
        obs, _, done, (mode, succ) = env.step(a)
        torch.cuda.synchronize()
        st, fl, sc = env.get_state()
        for e in (0, 63, 64, n - 1):
            np.testing.assert_allclose(st[:, e], g[name + "__states"][t + 1], atol=TOL, err_msg="%s step %d env %d" % (name, t, e))
        assert np.array_equal(obs[5].cpu().numpy(), g[name + "__obs"][t + 1])
        assert bool(done[5]) == bool(g[name + "__done"][t + 1]) and bool(succ[5]) == bool(g[name + "__succ"][t + 1])
        code = int((g[name + "__mode"][t + 1].astype(np.int64) * (1 << np.arange(9))).sum())
        assert int(mode[5]) == code
        assert (sc == t + 1).all() and not (fl & (1 << 16)).any()
        assert (st == st[:, :1]).all()        # identical inputs -> bit-identical lanes
    env.close()


def test_random_policy_rollout_vs_oracle(lib_loaded, init_qpos):
    """Device-side random policy (Philox) on 256 envs for a full episode with auto-reset; 6 envs are replayed on the
    oracle with the actions read back from the device."""
    from oracle.oracle import Oracle
    n, T = 256, 260
    env = _env(n)
    env.set_init_qpos(init_qpos)
    env.reset(); env.policy_begin()
    actions = torch.zeros(n, 7, dtype=torch.float64, device=env.device)
    picks = [0, 1, 77, 128, 200, 255]
    orcs = []
    for e in picks:
        o = Oracle(env.blob); o.env_start(init_qpos); o.env_reset(); orcs.append(o)
    n_done = 0
    for t in range(T):
        env.policy_action(42, 0, t, actions)
        obs, _, done, (mode, succ) = env.step(actions)
        torch.cuda.synchronize()
        a_host = actions.cpu().numpy()
        st, fl, sc = env.get_state()
        d_host = done.cpu().numpy().astype(bool)
        for o, e in zip(orcs, picks):
            ob, dn, md, su = o.env_step(a_host[e])
            so, fo = o.env_state()
            np.testing.assert_allclose(st[:, e], so, atol=TOL, err_msg="t=%d env=%d" % (t, e))
            assert dn == d_host[e] and np.array_equal(ob, obs[e].cpu().numpy())
            assert int(mode[e]) == int((md.astype(np.int64) * (1 << np.arange(9))).sum()) and bool(succ[e]) == su
            if dn:
                o.env_reset()
        n_done += int(d_host.sum())
        if d_host.any():                      # auto-reset finished envs, re-latch the harness' desired pose
            m = done.clone()                  # reset clears the done flags of the envs it resets: keep the mask
            env.reset(m); env.policy_begin(m)
            torch.cuda.synchronize()
            st2, fl2, sc2 = env.get_state()
            assert (sc2[d_host] == 0).all() and (sc2[~d_host] == sc[~d_host]).all()
            assert np.array_equal(st2[:, ~d_host], st[:, ~d_host])     # masked reset leaves the others untouched
    assert n_done >= n        # every env finished at least once (250-step cap)
    env.close()


def test_full_size_properties(lib_loaded, init_qpos):
    """BASELINE size (4096 envs): size-independent properties - determinism, batch-composition invariance,
    fast-path == eigen-path, state get/set round trip, sanity of outputs."""
    n, T = 4096, 12

    def run(n_envs, offset, fast=1, keep=None):
        env = _env(n_envs)
        env.set_option("ik_fast_path", fast)
        env.set_init_qpos(init_qpos)
        env.reset(); env.policy_begin()
        a = torch.zeros(n_envs, 7, dtype=torch.float64, device=env.device)
        for t in range(T):
            env.policy_action(42, offset, t, a)
            env.step(a)
        torch.cuda.synchronize()
        out = env.get_state() + (env.obs.cpu().numpy().copy(), env.done.cpu().numpy().copy())
        if keep is not None:
            keep.append(env)
        else:
            env.close()
        return out

    keep = []
    s1, f1, c1, o1, d1 = run(n, 0, keep=keep)
    s2, f2, c2, o2, d2 = run(n, 0)
    assert np.array_equal(s1, s2) and np.array_equal(f1, f2) and np.array_equal(o1, o2)          # deterministic
    s3, f3, c3, o3, d3 = run(1024, 2048)                                                          # shard of the same global envs
    assert np.array_equal(s3, s1[:, 2048:3072]) and np.array_equal(f3, f1[2048:3072])
    s4, f4, c4, o4, d4 = run(n, 0, fast=0)
    np.testing.assert_allclose(s4, s1, atol=1e-9)
    assert np.isfinite(s1).all() and (c1 == T).all() and not (f1 & (1 << 16)).any()
    assert len(np.unique(s1[25])) > 4000                                                           # envs really differ
    env = keep[0]
    env.set_state(s2, f2, c2)
    s5, f5, c5 = env.get_state()
    assert np.array_equal(s5, s2) and np.array_equal(f5, f2) and np.array_equal(c5, c2)
    counts = env.count_metrics().cpu().numpy()
    assert counts[0] == int(d1.sum()) and counts[1] == int(((f1 >> 13) & 1).sum()) and counts[2:].sum() == counts[1]
    env.close()


def test_sim_harness_end_to_end(lib_loaded):
    """Avoiding_Sim.test_agent with a scripted batch agent: straight line through the left gap -> every rollout
    succeeds with the same mode code; metrics follow the reference formulas."""
    from d3il_amd.simulation.avoiding_sim import Avoiding_Sim

    class LineAgent:
        def reset(self):
            pass

        def predict_batch(self, obs4):
            d = torch.zeros(obs4.shape[0], 2, dtype=torch.float64, device=obs4.device)
            d[:, 0], d[:, 1] = -0.002, 0.004
            return d

    sim = Avoiding_Sim(seed=0, device="cuda:0", render=False, n_cores=1, n_trajectories=96)
    successes, entropy = sim.test_agent(LineAgent())
    assert successes.shape == (96,) and float(successes.mean()) == 1.0
    assert entropy == 0.0                      # a single mode -> zero entropy
    g = np.load(os.path.join(G, "oracle_avoiding_rollout.npz"))
    code = int((g["succeed__mode"][-1].astype(np.int64) * (1 << np.arange(9))).sum())
    assert (sim.last_rollout["mode_code"].cpu().numpy() == code).all()
    assert int(sim.last_rollout["n_pos"][0]) == len(g["succeed__actions"]) + 1


def test_all_waves_identical_at_full_size(lib_loaded):
    """4096 envs (64 waves) fed identical actions: every lane of every wave must reproduce the golden rollout bit for
    bit equal to lane 0 - guards against lane/wave dependent corruption (register spilling hazards)."""
    g = np.load(os.path.join(G, "oracle_avoiding_rollout.npz"))
    n = 4096
    env = _env(n)
    env.set_init_qpos(g["init_qpos"])
    for name in ("collide", "zigzag"):
        env.reset()
        acts = g[name + "__actions"]
        for t in range(len(acts)):
            a = torch.as_tensor(np.tile(acts[t], (n, 1)), dtype=torch.float64, device=env.device).contiguous()
            env.step(a)
            if t % 8 == 7 or t == len(acts) - 1:
                torch.cuda.synchronize()
                st, fl, sc = env.get_state()
                assert (st == st[:, :1]).all() and (fl == fl[0]).all(), "%s step %d" % (name, t)
                np.testing.assert_allclose(st[:, 0], g[name + "__states"][t + 1], atol=TOL)
    env.close()


@pytest.mark.parametrize("n", [1, 63, 65, 1000])
def test_ragged_batch_sizes_and_masks(lib_loaded, init_qpos, n):
    """n_envs not a multiple of the wave size; masked reset of a subset; step cap; per-env results independent of n."""
    g = np.load(os.path.join(G, "oracle_avoiding_rollout.npz"))
    env = _env(n, max_steps_per_episode=20)
    env.set_init_qpos(g["init_qpos"])
    env.reset()
    acts = g["succeed__actions"]
    for t in range(25):
        a = torch.as_tensor(np.tile(acts[t], (n, 1)), dtype=torch.float64, device=env.device).contiguous()
        obs, _, done, (mode, succ) = env.step(a)
        torch.cuda.synchronize()
        st, fl, sc = env.get_state()
        np.testing.assert_allclose(st[:, n - 1], g["succeed__states"][t + 1], atol=TOL)
        assert bool(done.all()) == (t >= 19) and bool(done.any()) == (t >= 19)      # counter >= max_steps - 1
    mask = torch.zeros(n, dtype=torch.uint8, device=env.device)
    mask[::2] = 1
    env.reset(mask)
    torch.cuda.synchronize()
    st2, fl2, sc2 = env.get_state()
    m = mask.cpu().numpy().astype(bool)
    assert (sc2[m] == 0).all() and (sc2[~m] == 25).all()
    np.testing.assert_allclose(st2[:, 0], g["succeed__states"][0], atol=TOL)
    if n > 1:
        assert np.array_equal(st2[:, 1], st[:, 1])
    env.close()


def test_api_errors(lib_loaded):
    import ctypes as C
    from d3il_amd import capi
    from d3il_amd.model import blob
    L = capi.load()
    b = blob.load("avoiding")
    h = C.c_void_p()
    assert L.d3il_create(0, 0, 0, C.byref(b), C.sizeof(b), C.byref(h)) == -1          # n_envs <= 0
    assert L.d3il_create(0, 8, 0, C.byref(b), 17, C.byref(h)) == -2                     # blob size
    assert L.d3il_create(1, 8, 0, C.byref(b), C.sizeof(b), C.byref(h)) == -1           # task id does not match the blob
    b3 = blob.load("avoiding"); b3.task_id = 7
    assert L.d3il_create(7, 8, 0, C.byref(b3), C.sizeof(b3), C.byref(h)) == -5          # unknown task id
    b5 = blob.load("avoiding"); b5.task_id = 3
    assert L.d3il_create(3, 8, 0, C.byref(b5), C.sizeof(b5), C.byref(h)) == -2          # a Stacking blob must carry three boxes and the finger hull
    b4 = blob.load("avoiding"); b4.task_id = 2
    assert L.d3il_create(2, 8, 0, C.byref(b4), C.sizeof(b4), C.byref(h)) == -2 and b"task objects" in L.d3il_last_error()   # a Sorting blob without cubes
    b2 = blob.load("avoiding"); b2.body_mass[35] *= 1.01
    assert L.d3il_create(0, 8, 0, C.byref(b2), C.sizeof(b2), C.byref(h)) == -5 and b"specialised" in L.d3il_last_error()
    assert L.d3il_create(0, 8, 0, C.byref(b), C.sizeof(b), C.byref(h)) == 0
    assert L.d3il_reset(h, None, None, None) == -6 and b"d3il_start" in L.d3il_last_error()   # env.start() first
    assert L.d3il_destroy(h) == 0


def test_arbitrary_states_incl_arm_joint_limits(lib_loaded, init_qpos):
    """set_state -> step from mid-episode states (some with arm joints beyond their range: the rare general constraint
    path, some in rod contact), each env checked against the oracle started from the same state."""
    from oracle.oracle import Oracle
    g = np.load(os.path.join(G, "oracle_avoiding_rollout.npz"))
    rng = np.random.default_rng(5)
    n = 96
    env = _env(n)
    env.set_init_qpos(init_qpos)
    env.reset()
    torch.cuda.synchronize()
    states = np.zeros((42, n)); flags = np.zeros(n, dtype=np.uint32); steps = np.zeros(n, dtype=np.int32); acts = np.zeros((n, 7))
    names = ["random", "collide", "succeed", "zigzag"]
    for e in range(n):
        name = names[e % 4]
        T = len(g[name + "__actions"])
        t = int(rng.integers(1, T - 3)) if e % 8 else T - 3          # every 8th env: just before the episode ends
        s = g[name + "__states"][t].copy()
        if e % 3 == 0:                                               # push one arm joint beyond its MJCF range
            idx, val = [(5, 3.83), (3, 0.09), (1, -1.84), (0, 2.97)][(e // 3) % 4]
            s[idx] = val; s[9 + idx] = rng.normal(scale=0.3)
            s[28 + idx] = min(max(val, env.blob.ctrl_qmin[idx]), env.blob.ctrl_qmax[idx])
        else:
            s[9:18] += rng.normal(scale=0.02, size=9)
        f = g[name + "__flags"][t]
        mode = int((g[name + "__mode"][t].astype(np.int64) * (1 << np.arange(9))).sum())
        flags[e] = mode | (int(f[4]) << 9) | (int(f[5]) << 10) | (int(f[6]) << 11) | (int(f[1]) << 12) | (int(f[2]) << 13) | (int(f[3]) << 14) | (1 << 15)
        states[:, e] = s; steps[e] = t; acts[e] = g[name + "__actions"][t]
    env.set_state(states, flags, steps)
    a = torch.as_tensor(acts, dtype=torch.float64, device=env.device).contiguous()
    orcs = []
    for e in range(n):
        o = Oracle(env.blob); o.env_start(init_qpos); o.env_reset(); o.env_set_state(states[:, e], int(flags[e]), int(steps[e])); orcs.append(o)
    for k in range(2):
        obs, _, done, (mode, succ) = env.step(a)
        torch.cuda.synchronize()
        st, fl, sc = env.get_state()
        for e, o in enumerate(orcs):
            ob, dn, md, su = o.env_step(acts[e])
            so, fo = o.env_state()
            np.testing.assert_allclose(st[:, e], so, atol=TOL, err_msg="env %d pass %d" % (e, k))
            assert dn == bool(done[e]) and su == bool(succ[e]) and np.array_equal(ob, obs[e].cpu().numpy())
            assert int(mode[e]) == int((md.astype(np.int64) * (1 << np.arange(9))).sum())
        assert not (fl & (1 << 16)).any()
    env.close()


def test_auto_reset_equals_masked_reset(lib_loaded, init_qpos):
    """d3il_auto_reset == d3il_reset(mask=done) + d3il_policy_begin(mask=done), plus exact episode counters."""
    n, T = 512, 270
    a_env, b_env = _env(n), _env(n)
    for env in (a_env, b_env):
        env.set_init_qpos(init_qpos); env.reset(); env.policy_begin()
    act_a = torch.zeros(n, 7, dtype=torch.float64, device=a_env.device)
    act_b = torch.zeros_like(act_a)
    counts = torch.zeros(2, dtype=torch.int64, device=a_env.device)
    n_done = n_succ = 0
    for t in range(T):
        a_env.policy_action(7, 100, t, act_a); b_env.policy_action(7, 100, t, act_b)
        _, _, done_a, (_, succ_a) = a_env.step(act_a)
        _, _, done_b, (_, succ_b) = b_env.step(act_b)
        torch.cuda.synchronize()
        n_done += int(done_b.sum()); n_succ += int((succ_b * done_b).sum())
        a_env.auto_reset(counts)
        m = done_b.clone()                    # reset clears the done flags of the envs it resets: keep the mask
        b_env.reset(m); b_env.policy_begin(m)
    torch.cuda.synchronize()
    sa, fa, ca = a_env.get_state(); sb, fb, cb = b_env.get_state()
    assert np.array_equal(sa, sb) and np.array_equal(fa, fb) and np.array_equal(ca, cb)
    assert torch.equal(a_env.policy_des, b_env.policy_des)
    assert counts.tolist() == [n_done, n_succ] and n_done >= n
    a_env.close(); b_env.close()
