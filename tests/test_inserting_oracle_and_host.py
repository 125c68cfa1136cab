"""Inserting task (gate_insertion.py:157-514), CPU side: converter output of the gate scene, task logic pinned against the reference's
own Python (goldens made by tests/golden/gen_reference_goldens.py::gen_inserting_task from Gate_Insertion_Env), the CPU oracle env (resting heights, a
cube pushed into the walls, the rod itself on a wall) and the HOST build of the generic engine (d3il_amd/csrc/gen_step.h: the code the HIP kernels run)
following the oracle through those contacts."""
import os

import numpy as np
import pytest

from d3il_amd.model import blob as blob_mod
from oracle.oracle import InsertLogic, Oracle
from tests.hostcheck.hostcheck import GenHostCheck

HERE = os.path.dirname(os.path.abspath(__file__))
TARGETS = [[0.3575, 0.276, 0.0], [0.525, 0.4535, 0.0], [0.6925, 0.276, 0.0]]


def _quat(yaw):
    return [np.cos(yaw / 2), 0, 0, np.sin(yaw / 2)]


CTX = np.array([[0.42, -0.17, 0.0] + _quat(0.3), [0.6, -0.08, 0.0] + _quat(-0.5), [0.45, 0.02, 0.0] + _quat(0.1)])
WAY = [(0.45, -0.1), (0.45, 0.12), (0.40, 0.16), (0.40, 0.25)]      # through cube 3 (blue) and on into the left gate: the rod ends up on maze_3 / maze_5


@pytest.fixture(scope="module")
def ins_blob():
    return blob_mod.load("inserting")


@pytest.fixture(scope="module")
def js():
    return blob_mod.load_json("inserting")


@pytest.fixture(scope="module")
def init_qpos():
    # gate_insertion_objects.py:5: the start pose of Avoiding / Pushing
    return np.load(os.path.join(HERE, "golden", "ref_offline_ik.npz"))["avoiding__traj_last"].copy()


def test_converter_builds_the_gate_scene(js, ins_blob):
    names = [b["name"] for b in js["bodies"]]
    i0 = names.index("push_box1")
    assert names[i0:i0 + 6] == ["push_box1", "push_box2", "push_box3", "target_box1", "target_box2", "target_box3"]
    assert names[i0 + 6:i0 + 23] == ["maze_%d" % i for i in range(3, 20)] and "maze_1" not in names and "maze_2" not in names      # gate_insertion.py:236-237
    assert len(js["bodies"]) == 64 and ins_blob.n_obj == 3 and ins_blob.max_steps == 2000 and ins_blob.n_substeps == 35
    geoms = {js["bodies"][g["body"]]["name"]: g for g in js["geoms"] if js["bodies"][g["body"]]["name"].startswith(("push_box", "target_box", "maze_"))}
    for k in (1, 2, 3):
        assert geoms["push_box%d" % k]["size"] == [0.025, 0.025, 0.025] and abs(js["bodies"][names.index("push_box%d" % k)]["mass"] - 0.05) < 1e-15
        assert geoms["target_box%d" % k]["contype"] == 0 and geoms["target_box%d" % k]["conaffinity"] == 0 and not js["bodies"][names.index("target_box%d" % k)]["joints"]
    # the four diagonal walls carry the reference's un-normalised quaternions (0, 0.5, +-1, 0), normalised by the compiler
    for k, sgn in ((3, 1), (4, -1), (15, 1), (16, -1)):
        q = np.array(js["bodies"][names.index("maze_%d" % k)]["quat"])
        np.testing.assert_allclose(q, np.array([0, 0.5, sgn, 0]) / np.sqrt(1.25), atol=1e-15)
    np.testing.assert_allclose(np.array(ins_blob.task_f[:10]), np.array(TARGETS).reshape(-1).tolist() + [0.01], atol=0)
    o = Oracle(ins_blob)
    assert (o.nq, o.nv) == (30, 27)


def test_task_logic_matches_the_reference_code():
    g = np.load(os.path.join(HERE, "golden", "ref_inserting_task.npz"))
    E, T = g["succ"].shape
    assert sorted(np.unique(g["code"]).tolist()) == [0, 1, 2, 3, 4, 5, 6] and g["succ"].sum() > 100
    for e in range(E):
        L = InsertLogic(TARGETS, 0.01)
        for t in range(T):
            obs, succ, md, nm, code = L.step(g["box"][e, t], g["rob"][e, t])
            assert np.array_equal(obs, g["obs"][e, t]) and succ == bool(g["succ"][e, t]) and nm == g["n_mode"][e, t] and code == g["code"][e, t]
            assert abs(md - g["mean_distance"][e, t]) < 1e-15


def _host_err(h, o):
    qp, qv = o.state()
    e = ev = 0.0
    for k in range(3):
        p, q, v = h.box(k)
        e = max(e, np.abs(p - qp[7 * k:7 * k + 3]).max(), np.abs(q - qp[7 * k + 3:7 * k + 7]).max())
        ev = max(ev, np.abs(v - qv[6 * k:6 * k + 6]).max())
    return max(e, np.abs(h.s[:9] - qp[21:30]).max()), max(ev, np.abs(h.s[9:18] - qv[18:27]).max())


def test_oracle_and_host_engine_through_cube_and_rod_wall_contacts(ins_blob, init_qpos):
    """The rod pushes the blue cube north into the left gate and then meets the walls itself.  Oracle: resting height of a 5 cm cube on the table's own
    contact parameters, the cube really moves, rod <-> wall pairs appear in its contact list.  Host engine (statics: 19 inside the table + 8 frame beams;
    110 state rows): bit-equal reset observation, same trajectory to 1e-6 until the rod starts chattering on the wall, same flags-free run."""
    o = Oracle(ins_blob)
    o.env_start(init_qpos)
    h = GenHostCheck(ins_blob)
    assert (h.nb, h.ns, h.ns_core, h.n) == (3, 27, 19, 110)
    oo, oh = o.ins_reset(CTX), h.reset(init_qpos, CTX)
    assert oh.shape == (11,) and np.array_equal(oo, oh)
    assert _host_err(h, o)[0] < 1e-15
    z = float(o.body(ins_blob.tcp_body)[0][2])
    des = oo[:2].astype(float)
    wi, rod_wall_at, worst_before = 0, None, 0.0
    cubes = {g for g in range(ins_blob.ngeom) if ins_blob.geom_body[g] in [ins_blob.obj_body[k] for k in range(3)]}
    for t in range(100):
        d = np.array(WAY[wi]) - des
        n = np.linalg.norm(d)
        if n < 0.006 and wi < len(WAY) - 1:
            wi += 1
        des = des + (d / n * min(n, 0.006) if n > 0 else 0)
        a = np.array([des[0], des[1], z, 0, 1, 0, 0])
        oo, do, io = o.ins_step(a)
        oh, dh, ih = h.step(a, fast=bool(t % 2))
        con = o.contacts()
        on_wall = any(int(b) == ins_blob.rod_geom and int(a_) not in cubes for a_, b in con[:, 8:10]) if len(con) else False
        if on_wall and rod_wall_at is None:
            rod_wall_at = t
        e, ev = _host_err(h, o)
        assert e < 1e-6 and ev < 1e-3, (t, e, ev)
        assert do == dh and (io["mode"] | (io["n_mode"] << 3)) == ih["mode"] and io["success"] == ih["success"] and not (ih["flags"] & 0x1F0000)
        if rod_wall_at is None:
            worst_before = max(worst_before, e)
        if t == 5:      # at rest: the cube has been pushed out of the table (context z = 0) and sits ~6 mm above the body-frame origin plane
            assert abs(h.box(0)[0][2] - 0.005975) < 1e-5 and abs(o.state()[0][2] - h.box(0)[0][2]) < 1e-12
    assert rod_wall_at is not None and 70 <= rod_wall_at <= 95          # the rod reaches maze_3 after pushing the cube aside
    assert worst_before < 2e-7
    qp, _ = o.state()
    assert np.linalg.norm(qp[14:16] - CTX[2, :2]) > 0.1                   # the blue cube travelled > 10 cm
    assert abs(h.s[108 + 1] - io["mean_distance"]) < 1e-6 and h.s[108] == 0     # task words: mean_distance of the last step, no letters


def test_host_engine_mode_and_success_events(ins_blob, init_qpos):
    """Cubes put on their goals in the order green, blue, red: number of letters, mode_dict code 'gbr' = 4, success and the done of the NEXT step."""
    o = Oracle(ins_blob)
    o.env_start(init_qpos)
    h = GenHostCheck(ins_blob)
    ctx = CTX.copy()
    o.ins_reset(ctx)
    oh = h.reset(init_qpos, ctx)
    z = float(o.body(ins_blob.tcp_body)[0][2])
    a = np.array([oh[0], oh[1], z, 0, 1, 0, 0], float)
    for t in range(3):
        o.ins_step(a); h.step(a)
    rest_z = h.box(0)[0][2]
    seen = []
    for n, k in enumerate([1, 2, 0]):
        s = h.s.copy()
        r = 42 + 13 * k
        s[r:r + 3] = [TARGETS[k][0] + 0.002, TARGETS[k][1] - 0.001, rest_z]
        s[r + 3:r + 7] = [1, 0, 0, 0]
        s[r + 7:r + 13] = 0
        h.s[:] = s
        o.ins_set_state(s, int(h.f[0]), int(h.f[1]))
        for t in range(2):
            oo, do, io = o.ins_step(a)
            oh, dh, ih = h.step(a)
            assert np.array_equal(oo, oh) and do == dh and ih["mode"] == (io["mode"] | (io["n_mode"] << 3)) and ih["success"] == io["success"]
            seen.append((io["n_mode"], io["mode"], io["success"], do))
    assert seen == [(1, 0, False, False), (1, 0, False, False), (2, 0, False, False), (2, 0, False, False), (3, 4, True, True), (3, 4, True, True)]
    assert int(h.s[108]) == 3 | (2 << 2) | (3 << 4) | (1 << 6)
