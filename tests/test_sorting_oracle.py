"""Sorting-4 in the oracle (groundwork for the next SURVEY 8 row; no kernel yet): scene conversion, the reset transient the
reference produces (cubes are placed at z = 0.05 INSIDE the platform box and are pushed out through its top), rest state
against the closed form, and an end-to-end scripted push with the completion-order mode code."""
import numpy as np
import pytest

from oracle.oracle import Oracle


@pytest.fixture(scope="module")
def sort_blob():
    from d3il_amd.model import blob
    return blob.load("sorting")


@pytest.fixture(scope="module")
def sort_init_qpos():
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_offline_ik.npz"))
    return g["sorting__traj_last"].copy()


CTX = np.array([[0.42, -0.12, 0.05, 1, 0, 0, 0], [0.47, -0.02, 0.05, 1, 0, 0, 0],
                [0.6, -0.13, 0.05, 1, 0, 0, 0], [0.62, -0.03, 0.05, 1, 0, 0, 0]], float)


def test_scene(sort_blob):
    from d3il_amd.model import blob
    js = blob.load_json("sorting")
    o = Oracle(sort_blob)
    assert (o.nq, o.nv) == (37, 33) and sort_blob.n_obj == 4
    names = [b["name"] for b in js["bodies"]]
    assert names[24:28] == ["red_1", "red_2", "blue_1", "blue_2"] and names[28:36] == ["target_box_%d" % i for i in range(1, 9)]
    plat = [g for g in js["geoms"] if g["body"] == names.index("platform")][0]
    assert plat["size"] == [0.3, 0.3, 0.1] and plat["priority"] == 1 and plat["friction"][0] == 0.3


def test_reset_hop_and_rest_height(sort_blob, sort_init_qpos):
    o = Oracle(sort_blob)
    o.env_start(sort_init_qpos)
    obs = o.sort_reset(CTX)
    assert obs.shape == (14,)
    np.testing.assert_allclose(obs[[2, 3, 5, 6, 8, 9, 11, 12]], CTX[:, :2].reshape(-1), atol=1e-6)
    z = float(o.body(sort_blob.tcp_body)[0][2])
    des = obs[:2].astype(float)
    zs = []
    for t in range(30):
        obs, done, info = o.sort_step(np.concatenate([des, [z], [0, 1, 0, 0]]))
        qp, qv = o.state()
        zs.append(qp[[2, 9, 16, 23]].copy())
        assert info["mode"] == 240 and not info["success"] and not done          # np.packbits of four -1 entries
    zs = np.array(zs)
    # 8 cm inside the platform top, critically damped release (time constant 0.02 s): one hop of ~5 cm, then rest
    assert 0.17 < zs.max() < 0.20
    c = o.contacts()
    assert len(c) == 16 and abs(c[0, 7] - 0.3 / np.sqrt(3)) < 1e-12                # platform priority 1: its friction 0.3 is used
    # rest height: four corner contacts carry m g; contact parameters are the platform's (priority), closed form as for Pushing
    d0, dw, width, tc = 0.9, 0.95, 0.001, 0.02
    k = 1.0 / (dw * dw * tc * tc)
    r = 1e-5
    for _ in range(60):
        x = r / width
        y = 2.0 * x * x if x < 0.5 else 1 - 2.0 * (1 - x) ** 2
        d = d0 + y * (dw - d0)
        r = 9.81 * (1 - d) / (4 * k * d * d)
    for _ in range(25):
        o.sort_step(np.concatenate([des, [z], [0, 1, 0, 0]]))
    qp, qv = o.state()
    np.testing.assert_allclose(qp[[2, 9, 16, 23]] - 0.03, 0.1 - r, atol=2e-7)
    assert np.max(np.abs(qv[:24])) < 1e-5


def test_scripted_push_towards_the_bin_sets_the_mode_code(sort_blob, sort_init_qpos):
    o = Oracle(sort_blob)
    o.env_start(sort_init_qpos)
    obs = o.sort_reset(CTX)
    z = float(o.body(sort_blob.tcp_body)[0][2])
    des = obs[:2].astype(float)
    code = 240
    for t in range(170):
        box = obs[2:4].astype(float)
        if t < 12:
            target = des.copy()
        else:
            aligned = abs(des[0] - box[0]) < 0.008 and des[1] < box[1] - 0.02
            target = np.array([box[0], 0.36]) if aligned else box + np.array([0.0, -0.06])
        d = target - des
        n = np.linalg.norm(d)
        des = des + d / max(n, 1e-9) * min(0.006, n)
        obs, done, info = o.sort_step(np.concatenate([des, [z], [0, 1, 0, 0]]))
        assert o.solver_iter() < 60 and np.isfinite(obs).all()
        code = info["mode"]
        if code != 240:
            break
    qp, _ = o.state()
    assert code == 0b01110000                    # first completed box is a red one: mode[0] = 0, the others still -1
    assert 0.3 < qp[0] < 0.5 and qp[1] > 0.22    # red_1 crossed into the red bin's footprint
