"""The CPU oracle against golden vectors produced by the REFERENCE's own Python code
(tests/golden/gen_reference_goldens.py, shim-imported in the build container)."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("case", ["walk", "far", "limits", "unnorm"])
def test_ik_controller_matches_reference(oracle, case):
    g = np.load(os.path.join(G, "ref_ik_controller.npz"))
    oracle.ik_reset()
    K = g[case + "__control"].shape[0]
    for k in range(K):
        oracle.ik_setpoint(g[case + "__setpoint"][k])
        oracle.set_robot_state(g[case + "__jpos"][k], g[case + "__jvel"][k])
        tau = oracle.ik_control()
        oq, oqd = oracle.ik_state()
        # torques up to ~70 Nm; the reference solves with LAPACK SVD + LU, the oracle with Jacobi
        np.testing.assert_allclose(tau, g[case + "__control"][k], rtol=0, atol=1e-9)
        np.testing.assert_allclose(oq, g[case + "__old_q"][k], rtol=0, atol=1e-13)
        np.testing.assert_allclose(oqd, g[case + "__old_qd"][k], rtol=0, atol=1e-10)


def test_joint_pd_and_finger_controller_match_reference(oracle):
    g = np.load(os.path.join(G, "ref_pd_finger.npz"))
    for i in range(len(g["pd_q"])):
        oracle.set_robot_state(g["pd_q"][i], g["pd_v"][i])
        np.testing.assert_array_equal(oracle.pd_control(g["pd_qd"][i], g["pd_vd"][i]), g["pd_u"][i])
    for i in range(len(g["f_pos"])):
        oracle.set_robot_state(np.zeros(7), np.zeros(7), g["f_pos"][i], g["f_vel"][i], float(g["f_setw"][i]), bool(g["f_grasp"][i]))
        np.testing.assert_array_equal(oracle.finger_ctrl(), g["f_force"][i])
        # preprocessCommand: uff = [tau, ff] + bias (Robots.py:551-559)
        uff = np.concatenate([g["f_tau"][i], g["f_force"][i]]) + g["f_bias"][i]
        np.testing.assert_allclose(uff, g["f_uff"][i], rtol=0, atol=1e-13)


def test_avoiding_mode_logic_matches_reference(oracle):
    g = np.load(os.path.join(G, "ref_avoiding_task.npz"))
    for ep in range(g["paths"].shape[0]):
        for t in range(g["paths"].shape[1]):
            mode, succ = oracle.check_mode([g["paths"][ep, t, 0], g["paths"][ep, t, 1], 0.12], reset=(t == 0))
            assert np.array_equal(mode, g["modes"][ep, t]) and succ == bool(g["succ"][ep, t])


def test_metrics_match_reference():
    from d3il_amd.simulation.metrics import avoiding_metrics
    g = np.load(os.path.join(G, "ref_avoiding_task.npz"))
    enc, su = g["metric_enc"], g["metric_succ"]
    codes = (enc.astype(np.int64) * (1 << np.arange(9))).sum(1)
    hist = np.bincount(codes[su == 1], minlength=512)
    sr, ent = avoiding_metrics(len(su), int(su.sum()), hist)
    assert sr == pytest.approx(float(g["metric_success_rate"]), abs=1e-7)   # reference averages float32
    assert ent == pytest.approx(float(g["metric_entropy"]), abs=1e-12)


@pytest.mark.parametrize("task", ["avoiding", "sorting", "stacking"])
def test_offline_ik_matches_reference(avoiding_json, task):
    from d3il_amd.controllers.offline_ik import offline_ik
    from d3il_amd.kinematics import UrdfChain
    g = np.load(os.path.join(G, "ref_offline_ik.npz"))
    c = avoiding_json["controller"]
    q, iters, err = offline_ik(UrdfChain(avoiding_json["urdf_chain"]), c["default_qpos"], g[task + "__target"],
                               np.array(c["joint_pos_min"]), np.array(c["joint_pos_max"]))
    assert iters == int(g[task + "__iters"])
    np.testing.assert_allclose(q, g[task + "__traj_last"], rtol=0, atol=1e-12)
