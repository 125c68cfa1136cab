"""The kernels' per-environment math (d3il_amd/csrc/panda_step.h), compiled for the host by tests/hostcheck
(test-only, not the product path), against the oracle.  Two independent formulations: link-frame RNEA/CRBA
with packed LDL^T vs world-frame MuJoCo-style.  The GPU parity tests (-m gpu) repeat this through the C ABI."""
import numpy as np
import pytest


def test_constants_agree(hostcheck, oracle, avoiding_blob):
    dofw, rodw, masses, coms = hostcheck.consts()
    np.testing.assert_allclose(dofw, oracle.vec("dof_invweight0"), rtol=1e-12)
    rod_body = avoiding_blob.geom_body[avoiding_blob.rod_geom]
    assert rodw == pytest.approx(oracle.body(rod_body)[2][0], rel=1e-12)
    # fused link-7 group: link7 + link8 + hand + rod + rod:tip
    assert masses[6] == pytest.approx(0.417345 + 0.1 + 0.670782 + 0.09424777960769382 + 0.004188790204786391, rel=1e-14)


def test_dynamics_agree(hostcheck, oracle, avoiding_blob):
    rng = np.random.default_rng(0)
    for _ in range(25):
        q = np.concatenate([rng.uniform(-2, 2, 7), rng.uniform(0, 0.04, 2)])
        v = rng.standard_normal(9) * np.array([1] * 7 + [0.1, 0.1])
        oracle.set_state(q, v); oracle.forward()
        M, bias, tcp = hostcheck.dynamics(q, v)
        np.testing.assert_allclose(M, oracle.M(), atol=1e-13)
        np.testing.assert_allclose(bias, oracle.vec("qfrc_bias"), atol=1e-11)
        np.testing.assert_allclose(tcp, oracle.body(avoiding_blob.tcp_body)[0], atol=1e-14)
        pos, quat, J = hostcheck.ik_fk(q[:7])
        np.testing.assert_allclose(pos, oracle.fk(q[:7])[0], atol=1e-15)
        np.testing.assert_allclose(quat, oracle.fk(q[:7])[1], atol=1e-15)
        np.testing.assert_allclose(J, oracle.jac(q[:7]), atol=1e-15)


@pytest.mark.parametrize("fast", [False, True])
@pytest.mark.parametrize("case", ["walk", "far", "limits", "unnorm"])
def test_ik_controller_matches_reference_goldens(hostcheck, case, fast):
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_ik_controller.npz"))
    ikq, ikqd, flags = np.zeros(7), np.zeros(7), 0
    for k in range(g[case + "__control"].shape[0]):
        tau, flags = hostcheck.ik_control(g[case + "__setpoint"][k], g[case + "__jpos"][k], g[case + "__jvel"][k], ikq, ikqd, flags, fast)
        np.testing.assert_allclose(tau, g[case + "__control"][k], rtol=0, atol=1e-9)
        np.testing.assert_allclose(ikq, g[case + "__old_q"][k], rtol=0, atol=1e-13)


@pytest.mark.parametrize("fast", [False, True])
@pytest.mark.parametrize("policy", ["random", "collide", "succeed"])
def test_episode_matches_oracle(hostcheck, oracle, init_qpos, policy, fast):
    rng = np.random.default_rng(3)
    oracle.env_start(init_qpos)
    obs_o = oracle.env_reset()
    so, fo = oracle.env_state()
    s, f, obs = hostcheck.env_reset(init_qpos)
    np.testing.assert_allclose(s, so, atol=1e-11)
    assert np.array_equal(obs, obs_o)
    des = so[25:28].copy()
    fz = des[2]
    ended = None
    for t in range(250):
        d = {"random": rng.uniform(-0.01, 0.01, 2), "collide": np.array([0.0005, 0.004]), "succeed": np.array([-0.002, 0.004])}[policy]
        des[:2] += d
        a = np.array([des[0], des[1], fz, 0, 1, 0, 0])
        obs_o, done_o, mode_o, succ_o = oracle.env_step(a)
        so, fo = oracle.env_state()
        obs, done = hostcheck.env_step(s, f, a, fast)
        np.testing.assert_allclose(s, so, atol=1e-10)
        mode = np.array([(f[0] >> i) & 1 for i in range(9)])
        assert done == done_o and np.array_equal(mode, mode_o.astype(int)) and bool(f[0] & (1 << 13)) == succ_o
        assert np.array_equal(obs, obs_o) and f[1] == fo[0] and not (f[0] & (1 << 16))
        if done:
            ended = t
            break
    if policy == "collide":
        assert ended is not None and (f[0] & (1 << 14)) and not succ_o      # rod hit an obstacle
    if policy == "succeed":
        assert ended is not None and succ_o and bin(int(f[0]) & 0x1FF).count("1") == 3
    if policy == "random":
        assert ended == 249                                                   # step cap (max_steps - 1)


def _limit_cases(blob, g):
    s0 = g["random__states"][40].copy()
    out = []
    for idx, val, vel in ((5, 3.83, 0.5), (3, 0.09, 0.3), (1, -1.84, -0.2), (0, 2.97, 0.1)):
        s = s0.copy()
        s[idx], s[9 + idx] = val, vel
        s[28 + idx] = min(max(val, blob.ctrl_qmin[idx]), blob.ctrl_qmax[idx])
        out.append(s)
    return out


def test_arm_joint_limit_path_matches_oracle(hostcheck, oracle, avoiding_blob, init_qpos):
    """Arm joints pushed beyond their MJCF range: the rare 9-dof Newton path of the kernels vs the oracle's rows."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_avoiding_rollout.npz"))
    oracle.env_start(init_qpos); oracle.env_reset()
    for s in _limit_cases(avoiding_blob, g):
        fl = 1 << 15
        oracle.env_set_state(s, fl, 40)
        sh, f = s.copy(), np.array([fl, 40], dtype=np.int32)
        for t in range(3):
            a = g["random__actions"][40 + t]
            oracle.env_step(a)
            so, fo = oracle.env_state()
            hostcheck.env_step(sh, f, a, True)
            np.testing.assert_allclose(sh, so, atol=1e-9)
            assert not (f[0] & (1 << 16))
