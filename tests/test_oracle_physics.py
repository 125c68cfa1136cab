"""Known-answer and consistency tests of the oracle's MuJoCo-side restatement (parity there is unpinned:
no MuJoCo/pinocchio in this environment - these tests pin the physics against first principles instead)."""
import numpy as np
import pytest


def _quat2mat(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def test_urdf_fk_known_answer(oracle, avoiding_json):
    # SURVEY 8c KAT (i): default posture (MjRobot.py:200-211) -> TCP (0.550900, 0, 0.699822), quat (0, 0.996195, 0, 0.087156)
    pos, quat = oracle.fk(avoiding_json["controller"]["default_qpos"])
    np.testing.assert_allclose(pos, [0.550900, 0.0, 0.699822], atol=1e-6)
    np.testing.assert_allclose(quat, [0.0, 0.996195, 0.0, 0.087156], atol=1e-6)
    from d3il_amd.kinematics import UrdfChain
    ch = UrdfChain(avoiding_json["urdf_chain"])
    rng = np.random.default_rng(0)
    for _ in range(10):
        q = rng.uniform(-2, 2, 7)
        p2, q2 = ch.fk(q)
        p1, q1 = oracle.fk(q)
        np.testing.assert_allclose(p1, p2, atol=1e-14)
        np.testing.assert_allclose(q1, q2, atol=1e-14)
        np.testing.assert_allclose(oracle.jac(q), ch.jacobian(q), atol=1e-14)


def test_mjcf_and_urdf_chains_agree(oracle, avoiding_blob):
    # two independent data sources (panda_rod_invisible.xml vs panda_arm_hand_pinocchio.urdf) describe the same arm
    rng = np.random.default_rng(1)
    for _ in range(5):
        q = np.concatenate([rng.uniform(-2, 2, 7), [0.0, 0.0]])
        oracle.set_state(q, np.zeros(9))
        oracle.forward()
        tcp, _, _ = oracle.body(avoiding_blob.tcp_body)
        np.testing.assert_allclose(tcp, oracle.fk(q[:7])[0], atol=1e-6)


def test_bias_forces_satisfy_lagrange_equations(oracle, avoiding_json):
    """qfrc_bias == Mdot v - 1/2 d(v'Mv)/dq + dU/dq with M(q), U(q) differentiated numerically."""
    rng = np.random.default_rng(0)
    q = np.concatenate([np.array([-0.36, 0.42, -0.13, -2.05, 0.12, 2.46, 0.21]) + 0.1 * rng.standard_normal(7), [0.01, 0.02]])
    v = rng.standard_normal(9) * np.array([1, 1, 1, 1, 1, 1, 1, 0.05, 0.05])

    def M_of(qq):
        oracle.set_state(qq, np.zeros(9)); oracle.forward()
        return oracle.M()

    def U_of(qq):
        oracle.set_state(qq, np.zeros(9)); oracle.forward()
        tot = 0.0
        for i, bd in enumerate(avoiding_json["bodies"]):
            if bd["mass"] == 0:
                continue
            xpos, xquat, _ = oracle.body(i)
            tot += bd["mass"] * 9.81 * (xpos + _quat2mat(xquat) @ np.array(bd["ipos"]))[2]
        return tot

    M = M_of(q)
    assert np.abs(M - M.T).max() == 0 and np.linalg.eigvalsh(M).min() > 0
    eps = 1e-6
    I = np.eye(9)
    dM = [(M_of(q + eps * I[i]) - M_of(q - eps * I[i])) / (2 * eps) for i in range(9)]
    dU = np.array([(U_of(q + eps * I[i]) - U_of(q - eps * I[i])) / (2 * eps) for i in range(9)])
    lagr = sum(dM[i] * v[i] for i in range(9)) @ v - 0.5 * np.array([v @ dM[i] @ v for i in range(9)]) + dU
    oracle.set_state(q, v); oracle.forward()
    np.testing.assert_allclose(oracle.vec("qfrc_bias"), lagr, atol=5e-6)


def test_gravity_compensated_hold_has_zero_arm_acceleration(oracle, init_qpos):
    # SURVEY 8c KAT (iii): ctrl = qfrc_bias at rest => qacc = 0 for the arm
    q = np.concatenate([init_qpos, [0.02, 0.02]])
    oracle.set_state(q, np.zeros(9)); oracle.set_ctrl(np.zeros(9)); oracle.forward()
    bias = oracle.vec("qfrc_bias")
    oracle.set_ctrl(bias); oracle.forward()
    assert np.abs(oracle.vec("qacc")).max() < 1e-9


def test_actuator_force_clamp(oracle, init_qpos):
    q = np.concatenate([init_qpos, [0.02, 0.02]])
    oracle.set_state(q, np.zeros(9)); oracle.set_ctrl(np.array([1e3, -1e3, 50, 0, 100, -100, 5, 200, -200.0])); oracle.forward()
    np.testing.assert_array_equal(oracle.vec("qfrc_actuator"), [87, -87, 50, 0, 12, -12, 5, 70, -70])


def test_energy_is_conserved_without_actuation_and_damping(oracle, init_qpos, avoiding_json):
    """Free swing of the arm (fingers parked at mid-range, zero finger velocity => no damping loss): the
    semi-implicit Euler energy error over 200 steps of 1 ms stays O(h)."""
    def energy():
        qpos, qvel = oracle.state()
        oracle.forward()
        T = 0.5 * qvel @ oracle.M() @ qvel
        U = 0.0
        for i, bd in enumerate(avoiding_json["bodies"]):
            if bd["mass"] == 0:
                continue
            xpos, xquat, _ = oracle.body(i)
            U += bd["mass"] * 9.81 * (xpos + _quat2mat(xquat) @ np.array(bd["ipos"]))[2]
        return T + U
    q = np.concatenate([init_qpos, [0.02, 0.02]])
    oracle.set_state(q, np.zeros(9)); oracle.set_ctrl(np.zeros(9))
    e0 = energy()
    for _ in range(200):
        oracle.mj_step()
    qpos, qvel = oracle.state()
    assert np.abs(qvel[:7]).max() > 0.1          # it really moved
    assert abs(energy() - e0) < 2e-2 * max(1.0, abs(e0)) * 0.2


def test_joint_limit_soft_constraint(oracle, init_qpos):
    # finger pushed 1 mm beyond its upper limit: one unilateral row, force pushes back (negative on the dof)
    q = np.concatenate([init_qpos, [0.041, 0.02]])
    oracle.set_state(q, np.zeros(9)); oracle.set_ctrl(np.zeros(9)); oracle.forward()
    f, aref, D = oracle.efc()
    assert len(f) == 1 and f[0] > 0 and aref[0] > 0
    fc = oracle.vec("qfrc_constraint")
    assert fc[7] < 0 and abs(fc[7] + f[0]) < 1e-12
    # solimp 0.9 0.95 0.001 at |r| = width -> d = 0.95 ; R = (1-d)/d * invweight ; k = 1/(0.95^2 0.02^2), b = 2/(0.95 0.02)
    invw = oracle.vec("dof_invweight0")[7]
    assert D[0] == pytest.approx(1.0 / ((1 - 0.95) / 0.95 * invw), rel=1e-12)
    assert aref[0] == pytest.approx(1.0 / (0.95 ** 2 * 0.02 ** 2) * 0.95 * 0.001, rel=1e-12)


def test_rod_obstacle_contact(oracle, init_qpos, avoiding_blob):
    """Move the arm until the rod overlaps obstacle l1 laterally: one contact, normal horizontal and pointing
    from the obstacle to the rod, depth = r1 + r2 - axis distance, contact force repulsive."""
    from d3il_amd.kinematics import UrdfChain
    from d3il_amd.controllers.offline_ik import offline_ik
    from d3il_amd.model import blob
    js = blob.load_json("avoiding")
    c = js["controller"]
    # rod axis goes through the TCP; put the TCP 0.035 m from the l1 axis (r = 0.03 + 0.01 -> 5 mm penetration)
    tgt = [0.5 + 0.035, -0.1, 0.12, 0, 1, 0, 0]
    q7, _, err = offline_ik(UrdfChain(js["urdf_chain"]), init_qpos, tgt, np.array(c["joint_pos_min"]), np.array(c["joint_pos_max"]))
    oracle.set_state(np.concatenate([q7, [0.04, 0.04]]), np.zeros(9)); oracle.set_ctrl(np.zeros(9)); oracle.forward()
    con = oracle.contacts()
    assert len(con) == 1
    dist, pos, nrm, mu = con[0, 0], con[0, 1:4], con[0, 4:7], con[0, 7]
    assert int(con[0, 8]) == avoiding_blob.obst_geom[0] and int(con[0, 9]) == avoiding_blob.rod_geom
    assert dist == pytest.approx(-0.005, abs=2e-4)
    assert abs(nrm[2]) < 0.05 and nrm[0] > 0.99
    assert mu == pytest.approx(1.0 / np.sqrt(3.0), rel=1e-12)        # friction 1, impratio 3
    f, aref, D = oracle.efc()
    assert len(f) == 3 and f[0] > 0
    assert D[1] == pytest.approx(3 * D[0], rel=1e-12) and D[2] == pytest.approx(3 * D[0], rel=1e-12)
    assert oracle.solver_iter() >= 1
    # stationarity of the primal problem: M (qacc - qacc_smooth) = qfrc_constraint
    r = oracle.M() @ (oracle.vec("qacc") - oracle.vec("qacc_smooth")) - oracle.vec("qfrc_constraint")
    assert np.abs(r).max() < 1e-9


def test_unsupported_pairs_are_listed_not_silently_dropped(oracle):
    sup, uns = oracle.pairs(True), oracle.pairs(False)
    assert len(sup) == 12 and len(uns) > 0   # rod vs 6 obstacles + 6 (unreachable) table-foot cylinders
