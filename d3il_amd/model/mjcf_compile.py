"""MJCF-subset / URDF / gin -> flat model description ("model blob") converter.

Runs only where the reference *data* files are present (this container); its
output (``d3il_amd/model/blobs/*.json``) is committed, so nothing on the GPU
box ever reads ``/root/reference``.  No reference code is used: only the
parameters in the MJCF/URDF/gin data files and the object constants listed in
the task's ``*_objects.py`` (restated below with file:line citations).

What it restates of MuJoCo 2.3.2's model compiler [ext]:
  * default classes + ``childclass`` inheritance for geom/joint attributes,
  * body tree with pos/quat, hinge/slide/free joints, explicit ``<inertial>``
    or geom-derived inertia (density 1000, or ``mass=`` scaling),
  * quaternion normalisation at compile time,
  * motors with ``forcerange``, ``<contact><exclude>``.

Reference data consumed (file:line):
  environments/d3il/models/mj/surroundings/base.xml:1-20
  environments/d3il/models/mujoco/surroundings/lab_surrounding.xml:1-117
  environments/d3il/models/mj/robot/panda_rod_invisible.xml:1-141
  environments/d3il/models/common/robots/panda_arm_hand_pinocchio.urdf:51-336
  environments/d3il/d3il_sim/controllers/Config/mujoco_controller_config.gin:6-37
  environments/d3il/envs/gym_avoiding_env/gym_avoiding/envs/objects/avoiding_objects.py:5-63
  environments/d3il/d3il_sim/sims/mj_beta/MjPrimLoader.py:6-42 (how primitives become bodies)
  environments/d3il/d3il_sim/core/Robots.py:57-65 (controller-side joint limits)
"""
from __future__ import annotations

import ast
import json
import math
import os
import re
import xml.etree.ElementTree as ET

import numpy as np

REF = os.environ.get("D3IL_REFERENCE", "/root/reference")
D3IL = os.path.join(REF, "environments", "d3il")

GEOM_DEFAULTS = dict(
    type="sphere", contype=1, conaffinity=1, condim=3, priority=0,
    friction=[1.0, 0.005, 0.0001], margin=0.0, gap=0.0, solmix=1.0,
    solref=[0.02, 1.0], solimp=[0.9, 0.95, 0.001, 0.5, 2.0],
    density=1000.0, pos=[0.0, 0.0, 0.0], quat=[1.0, 0.0, 0.0, 0.0],
)
JOINT_DEFAULTS = dict(
    type="hinge", axis=[0.0, 0.0, 1.0], pos=[0.0, 0.0, 0.0], damping=0.0,
    limited=False, range=[0.0, 0.0], armature=0.0, frictionloss=0.0,
    solreflimit=[0.02, 1.0], solimplimit=[0.9, 0.95, 0.001, 0.5, 2.0], margin=0.0,
)


def _floats(s):
    return [float(x) for x in s.split()]


def _normalize(q):
    q = np.asarray(q, dtype=np.float64)
    n = math.sqrt(float(np.dot(q, q)))
    return (q / n).tolist()


def quat_mul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return [aw * bw - ax * bx - ay * by - az * bz,
            aw * bx + ax * bw + ay * bz - az * by,
            aw * by - ax * bz + ay * bw + az * bx,
            aw * bz + ax * by - ay * bx + az * bw]


def quat2mat(q):
    w, x, y, z = q
    return np.array([
        [w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z]])


class Defaults:
    """Nested <default class=...> table (only geom/joint are used by the task files)."""

    def __init__(self):
        self.classes = {"main": {"geom": {}, "joint": {}, "parent": None}}

    def parse(self, elem, parent="main"):
        for child in elem:
            if child.tag == "default":
                name = child.get("class")
                self.classes[name] = {"geom": {}, "joint": {}, "parent": parent}
                self.parse(child, name)
            elif child.tag in ("geom", "joint"):
                self.classes[parent][child.tag].update(child.attrib)

    def resolve(self, cls, tag):
        chain = []
        c = cls or "main"
        while c is not None:
            chain.append(self.classes[c][tag])
            c = self.classes[c]["parent"]
        out = {}
        for d in reversed(chain):
            out.update(d)
        return out


def _geom_attrs(elem, defaults, childclass):
    cls = elem.get("class") or childclass
    a = dict(defaults.resolve(cls, "geom"))
    a.update(elem.attrib)
    g = dict(GEOM_DEFAULTS)
    g["name"] = a.get("name", "")
    g["type"] = a.get("type", g["type"])
    for k in ("contype", "conaffinity", "condim", "priority"):
        if k in a:
            g[k] = int(a[k])
    for k in ("margin", "gap", "solmix", "density"):
        if k in a:
            g[k] = float(a[k])
    if "friction" in a:
        f = _floats(a["friction"])
        g["friction"] = f + GEOM_DEFAULTS["friction"][len(f):]
    if "solref" in a:
        g["solref"] = _floats(a["solref"])
    if "solimp" in a:
        s = _floats(a["solimp"])
        g["solimp"] = s + GEOM_DEFAULTS["solimp"][len(s):]
    if "pos" in a:
        g["pos"] = _floats(a["pos"])
    if "quat" in a:
        g["quat"] = _normalize(_floats(a["quat"]))
    g["size"] = _floats(a["size"]) if "size" in a else []
    g["mass"] = float(a["mass"]) if "mass" in a else None
    g["mesh"] = a.get("mesh")
    return g


def _joint_attrs(elem, defaults, childclass):
    if elem.tag == "freejoint":
        j = dict(JOINT_DEFAULTS)
        j.update(type="free", name=elem.get("name", ""))
        return j
    cls = elem.get("class") or childclass
    a = dict(defaults.resolve(cls, "joint"))
    a.update(elem.attrib)
    j = dict(JOINT_DEFAULTS)
    j["name"] = a.get("name", "")
    j["type"] = a.get("type", "hinge")
    if "axis" in a:
        j["axis"] = _normalize(_floats(a["axis"]))
    if "pos" in a:
        j["pos"] = _floats(a["pos"])
    if "damping" in a:
        j["damping"] = float(a["damping"])
    if "limited" in a:
        j["limited"] = a["limited"] == "true"
    if "range" in a:
        j["range"] = _floats(a["range"])
    return j


def geom_mass_inertia(g):
    """Mass and diagonal inertia (about the geom centre, geom axes) of a primitive.
    MuJoCo: default density 1000 kg/m^3, or the shape inertia scaled to ``mass=`` [ext]."""
    t, s = g["type"], g["size"]
    if t == "sphere":
        r = s[0]
        vol = 4.0 / 3.0 * math.pi * r ** 3
        unit = [0.4 * r * r] * 3
    elif t == "cylinder":
        r, h = s[0], s[1]
        vol = math.pi * r * r * 2 * h
        ixy = (3 * r * r + 4 * h * h) / 12.0
        unit = [ixy, ixy, r * r / 2.0]
    elif t == "box":
        x, y, z = s[:3]
        vol = 8 * x * y * z
        unit = [(y * y + z * z) / 3.0, (x * x + z * z) / 3.0, (x * x + y * y) / 3.0]
    else:
        raise ValueError("no analytic inertia for geom type %s" % t)
    mass = g["mass"] if g["mass"] is not None else vol * g["density"]
    return mass, [mass * u for u in unit]


class Model:
    def __init__(self):
        self.bodies = [dict(name="world", parent=-1, pos=[0.0] * 3, quat=[1.0, 0, 0, 0], mass=0.0,
                            ipos=[0.0] * 3, iquat=[1.0, 0, 0, 0], inertia=[0.0] * 3, joints=[],
                            explicit_inertial=True)]
        self.geoms = []
        self.excludes = []
        self.actuators = []
        self.defaults = Defaults()
        self.option = {}

    def add_body(self, elem, parent_id, childclass, rename=lambda s: s):
        cc = elem.get("childclass") or childclass
        b = dict(name=rename(elem.get("name", "")), parent=parent_id,
                 pos=_floats(elem.get("pos", "0 0 0")),
                 quat=_normalize(_floats(elem.get("quat", "1 0 0 0"))),
                 mass=0.0, ipos=[0.0] * 3, iquat=[1.0, 0, 0, 0], inertia=[0.0] * 3,
                 joints=[], explicit_inertial=False)
        bid = len(self.bodies)
        self.bodies.append(b)
        own_geoms = []
        for ch in elem:
            if ch.tag == "inertial":
                b["mass"] = float(ch.get("mass"))
                b["ipos"] = _floats(ch.get("pos", "0 0 0"))
                b["iquat"] = _normalize(_floats(ch.get("quat", "1 0 0 0")))
                b["inertia"] = _floats(ch.get("diaginertia"))
                b["explicit_inertial"] = True
            elif ch.tag in ("joint", "freejoint"):
                j = _joint_attrs(ch, self.defaults, cc)
                j["name"] = rename(j["name"])
                b["joints"].append(j)
            elif ch.tag == "geom":
                g = _geom_attrs(ch, self.defaults, cc)
                g["name"] = rename(g["name"])
                g["body"] = bid
                self.geoms.append(g)
                own_geoms.append(g)
        if not b["explicit_inertial"] and own_geoms:
            self._inertia_from_geoms(b, own_geoms)
        for ch in elem:
            if ch.tag == "body":
                self.add_body(ch, bid, cc, rename)
        return bid

    @staticmethod
    def _inertia_from_geoms(b, geoms):
        """Body inertial frame from its geoms (MuJoCo ``inertiafromgeom=auto``) [ext]."""
        masses, coms, tensors = [], [], []
        for g in geoms:
            if g["type"] in ("mesh", "plane"):
                continue
            m, diag = geom_mass_inertia(g)
            R = quat2mat(g["quat"])
            masses.append(m)
            coms.append(np.array(g["pos"]))
            tensors.append(R @ np.diag(diag) @ R.T)
        if not masses:
            return
        M = sum(masses)
        com = sum(m * c for m, c in zip(masses, coms)) / M
        I = np.zeros((3, 3))
        for m, c, T in zip(masses, coms, tensors):
            d = c - com
            I += T + m * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
        w, V = np.linalg.eigh(I)
        # principal axes; for the axis-aligned primitives used here V is a signed permutation.
        if np.allclose(I, np.diag(np.diag(I)), atol=1e-15):
            b["inertia"] = np.diag(I).tolist()
            b["iquat"] = [1.0, 0.0, 0.0, 0.0]
        else:  # pragma: no cover - not reached by the four task models
            if np.linalg.det(V) < 0:
                V[:, 2] = -V[:, 2]
            b["inertia"] = w.tolist()
            b["iquat"] = _normalize(mat2quat(V))
        b["mass"] = float(M)
        b["ipos"] = com.tolist()


def mat2quat(R):
    """Rotation matrix -> quaternion (w,x,y,z), Eigen's branch rule (used by pinocchio) [ext]."""
    t = R[0, 0] + R[1, 1] + R[2, 2]
    q = [0.0] * 4
    if t > 0:
        t = math.sqrt(t + 1.0)
        q[0] = 0.5 * t
        t = 0.5 / t
        q[1] = (R[2, 1] - R[1, 2]) * t
        q[2] = (R[0, 2] - R[2, 0]) * t
        q[3] = (R[1, 0] - R[0, 1]) * t
    else:
        i = 0
        if R[1, 1] > R[0, 0]:
            i = 1
        if R[2, 2] > R[i, i]:
            i = 2
        j = (i + 1) % 3
        k = (j + 1) % 3
        t = math.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
        q[1 + i] = 0.5 * t
        t = 0.5 / t
        q[0] = (R[k, j] - R[j, k]) * t
        q[1 + j] = (R[j, i] + R[i, j]) * t
        q[1 + k] = (R[k, i] + R[i, k]) * t
    return q


def prim_body(name, typ, pos, quat, size, mass=0.1, static=False, visual_only=False):
    """XML element for a primitive object, as MjPrimLoader.py:6-42 builds it."""
    body = ET.Element("body", name=name, pos=" ".join(map(str, pos)), quat=" ".join(map(str, quat)))
    geom = ET.SubElement(body, "geom", type=typ, name="%s:geom" % name, size=" ".join(map(str, size)))
    if mass:
        geom.set("mass", str(mass))
    if visual_only:
        geom.set("contype", "0")
        geom.set("conaffinity", "0")
    if not static:
        ET.SubElement(body, "freejoint")
    return body


def parse_urdf_chain(path, tip_frame="panda_grasptarget"):
    """Serial chain world -> tip of the pinocchio URDF: per joint the fixed placement
    (xyz, rpy -> R = Rz(y) Ry(p) Rx(r)) and, for revolute joints, the axis."""
    root = ET.parse(path).getroot()
    joints = {}
    for j in root.findall("joint"):
        o = j.find("origin")
        xyz = _floats(o.get("xyz", "0 0 0")) if o is not None else [0.0] * 3
        rpy = _floats(o.get("rpy", "0 0 0")) if o is not None else [0.0] * 3
        ax = j.find("axis")
        joints[j.find("child").get("link")] = dict(
            name=j.get("name"), type=j.get("type"), parent=j.find("parent").get("link"),
            xyz=xyz, rpy=rpy, axis=_floats(ax.get("xyz")) if ax is not None else [0.0, 0.0, 1.0])
    chain = []
    link = tip_frame
    while link in joints:
        chain.append(joints[link])
        link = joints[link]["parent"]
    chain.reverse()
    out = []
    for j in chain:
        r, p, y = j["rpy"]
        Rx = np.array([[1, 0, 0], [0, math.cos(r), -math.sin(r)], [0, math.sin(r), math.cos(r)]])
        Ry = np.array([[math.cos(p), 0, math.sin(p)], [0, 1, 0], [-math.sin(p), 0, math.cos(p)]])
        Rz = np.array([[math.cos(y), -math.sin(y), 0], [math.sin(y), math.cos(y), 0], [0, 0, 1]])
        out.append(dict(name=j["name"], type=j["type"], xyz=j["xyz"], rpy=j["rpy"],
                        R=(Rz @ Ry @ Rx).tolist(), axis=j["axis"]))
    return out


def parse_gin(path):
    cfg = {}
    pat = re.compile(r"^\s*[\w\.]+\.(\w+)\.(\w+)\s*=\s*(.+)$")
    for line in open(path):
        if line.lstrip().startswith("#"):
            continue
        m = pat.match(line)
        if m:
            cfg.setdefault(m.group(1), {})[m.group(2)] = ast.literal_eval(m.group(3).strip())
    return cfg


def build_scene(robot_xml, objects, task):
    """Assemble the scene the way mj_scene_parser.py:36-53 does: base + surrounding include
    + world-body objects + robot include (names get the ``_rb0`` infix, MjRobot.py:272-284)."""
    m = Model()
    base = ET.parse(os.path.join(D3IL, "models/mj/surroundings/base.xml")).getroot()
    opt = base.find("option").attrib
    m.option = dict(timestep=0.001, gravity=_floats(opt["gravity"]), impratio=float(opt["impratio"]),
                    tolerance=float(opt["tolerance"]), cone=opt["cone"], solver=opt["solver"],
                    iterations=100, ls_iterations=50, ls_tolerance=0.01, integrator="Euler")
    robot = ET.parse(os.path.join(D3IL, "models/mj/robot", robot_xml)).getroot()
    for d in robot.findall("default"):
        m.defaults.parse(d)

    def rb(s):  # MjRobot.add_id2model_key (MjRobot.py:272-284)
        if not s:
            return s
        parts = s.split("_")
        parts.insert(1, "rb0")
        return "_".join(parts)

    # the template's class names are renamed too; keep a lookup from the renamed to the parsed name
    for wb in base.findall("worldbody"):
        for b in wb.findall("body"):
            m.add_body(b, 0, None)
    surr = ET.parse(os.path.join(D3IL, "models/mujoco/surroundings/lab_surrounding.xml")).getroot()
    for wb in surr.findall("worldbody"):
        for b in wb.findall("body"):
            m.add_body(b, 0, None)
    for ob in objects:
        m.add_body(ob, 0, None)
    for wb in robot.findall("worldbody"):
        for b in wb.findall("body"):
            m.add_body(b, 0, None, rename=rb)
    for c in robot.findall("contact"):
        for e in c.findall("exclude"):
            m.excludes.append([rb(e.get("body1")), rb(e.get("body2"))])
    for a in robot.findall("actuator"):
        for mot in a.findall("motor"):
            m.actuators.append(dict(name=rb(mot.get("name")), joint=rb(mot.get("joint")),
                                    forcerange=_floats(mot.get("forcerange")),
                                    forcelimited=mot.get("forcelimited") == "true"))
    return m


def avoiding_objects():
    """avoiding_objects.py:5-63 restated as data (cylinder size = radius, half-length)."""
    mid, off, y0, dy = 0.5, 0.075, -0.1, 0.18
    objs = [
        prim_body("l1_obs", "cylinder", [mid, y0, 0], [1, 0, 0, 0], [0.03, 0.07], static=True),
        prim_body("l2_top_obs", "cylinder", [mid - off, y0 + dy, 0], [1, 0, 0, 0], [0.025, 0.1], static=True),
        prim_body("l2_bottom_obs", "cylinder", [mid + off, y0 + dy, 0], [1, 0, 0, 0], [0.025, 0.1], static=True),
        prim_body("l3_top_obs", "cylinder", [mid - 2 * off, y0 + 2 * dy, 0], [1, 0, 0, 0], [0.025, 0.1], static=True),
        prim_body("l3_mid_obs", "cylinder", [mid, y0 + 2 * dy, 0], [1, 0, 0, 0], [0.025, 0.1], static=True),
        prim_body("l3_bottom_obs", "cylinder", [mid + 2 * off, y0 + 2 * dy, 0], [1, 0, 0, 0], [0.025, 0.1], static=True),
        prim_body("finish_line", "box", [0.4, y0 + 2.5 * dy, 0], [1, 0, 0, 0], [0.5, 0.01, 0.005],
                  static=True, visual_only=True),
    ]
    return objs


def to_blob(m, task, task_const):
    gin = parse_gin(os.path.join(D3IL, "d3il_sim/controllers/Config/mujoco_controller_config.gin"))
    chain = parse_urdf_chain(os.path.join(D3IL, "models/common/robots/panda_arm_hand_pinocchio.urdf"))
    name2body = {b["name"]: i for i, b in enumerate(m.bodies)}
    blob = dict(
        version=1, task=task, option=m.option,
        bodies=[{k: v for k, v in b.items() if k != "explicit_inertial"} for b in m.bodies],
        geoms=[{k: v for k, v in g.items() if k not in ("density",)} for g in m.geoms],
        excludes=[[name2body[a], name2body[b]] for a, b in m.excludes],
        actuators=m.actuators,
        urdf_chain=chain,
        controller=dict(
            joint_pd=gin["JointPDGains"],
            cart_pos_quat=gin["CartPosQuatControllerConfig"],
            # core/Robots.py:57-65 (controller-side limits, differ from the MJCF ranges)
            joint_pos_min=[-2.8973, -1.7628, -2.8973, -3.0718, -2.8973, -0.0175, -2.8973],
            joint_pos_max=[2.8973, 1.7628, 2.0, -0.0698, 2.8973, 3.7525, 2.8973],
            # MjRobot.py:200-211
            default_qpos=[3.57795216e-09, 1.74532920e-01, 3.30500960e-08, -8.72664630e-01,
                          -1.14096181e-07, 1.22173047e00, 7.85398126e-01],
        ),
        task_const=task_const,
    )
    return blob


def build_avoiding():
    m = build_scene("panda_rod_invisible.xml", avoiding_objects(), "avoiding")
    tc = dict(
        n_substeps=35, max_steps=250,                        # avoiding.py:55-56
        init_end_eff_pos=[0.525, -0.28, 0.12], init_end_eff_quat=[0, 1, 0, 0],  # avoiding_objects.py:5, avoiding.py:141-152
        obstacles=["l1_obs", "l2_top_obs", "l2_bottom_obs", "l3_top_obs", "l3_mid_obs", "l3_bottom_obs"],
        rod_geom="rod:geom_rb0", tcp_body="tcp_rb0",        # mj_helper.py:15-17, MjRobot.py:138
        # avoiding.py:94-107
        l1_ypos=-0.1, l2_ypos=-0.1 + 0.18, l3_ypos=-0.1 + 2 * 0.18, goal_ypos=-0.1 + 2.5 * 0.18,
        l1_xpos=0.5, l2_top_xpos=0.5 - 0.075, l2_bottom_xpos=0.5 + 0.075,
        l3_top_xpos=0.5 - 2 * 0.075, l3_mid_xpos=0.5, l3_bottom_xpos=0.5 + 2 * 0.075,
    )
    return to_blob(m, "avoiding", tc)


def pushing_objects():
    """pushing_objects.py:5-61 restated as data, in the order pushing.py:227-235 adds them to the scene."""
    return [
        prim_body("push_box", "box", [0.4, -0.3, -0.0072], [0, 1, 0, 0], [0.03, 0.03, 0.03], mass=0.05),
        prim_body("push_box2", "box", [0.5, -0.3, -0.0072], [0, 1, 0, 0], [0.03, 0.03, 0.03], mass=0.05),
        prim_body("target_box_1", "box", [0.42, 0.3, 0], [0, 1, 0, 0], [0.05, 0.05, 0.04], static=True, visual_only=True),
        prim_body("target_box_2", "box", [0.63, 0.3, 0], [0, 1, 0, 0], [0.05, 0.05, 0.04], static=True, visual_only=True),
    ]


def build_pushing():
    m = build_scene("panda_rod_invisible.xml", pushing_objects(), "pushing")
    tc = dict(
        n_substeps=35, max_steps=400,                        # pushing.py:174-175
        init_end_eff_pos=[0.525, -0.28, 0.12], init_end_eff_quat=[0, 1, 0, 0],  # pushing_objects.py:5, pushing.py:305-317
        rod_geom="rod:geom_rb0", tcp_body="tcp_rb0",
        objects=["push_box", "push_box2"],
        target_pos1=[0.42, 0.3, 0.0], target_pos2=[0.63, 0.3, 0.0],           # pushing_objects.py:11-15
        target_min_dist=0.05,                                                   # pushing.py:251
    )
    return to_blob(m, "pushing", tc)


def sorting_objects(num_boxes=4):
    """sorting_objects.py:5-228 restated as data, in the order Sorting_Env builds its object list (sorting.py:208-215):
    red boxes, blue boxes, the eight bin walls, the platform (models/mj/common-objects/sorting/platform.xml with the pose
    SortingObject.mj_load overrides: pos 0.5 -0.1 0, quat 1 0 0 0)."""
    n = num_boxes // 2
    objs = []
    for colour in ("red", "blue"):
        for i in range(n):
            objs.append(prim_body("%s_%d" % (colour, i + 1), "box", [0.5, -0.1, 0.0], [0, 1, 0, 0], [0.03, 0.03, 0.03], mass=0.05))
    walls = [([0.4, 0.41, 0.0], [0.1, 0.01, 0.1]), ([0.3, 0.32, 0.0], [0.005, 0.1, 0.1]), ([0.5, 0.32, 0.0], [0.005, 0.1, 0.1]),
             ([0.4, 0.22, 0.0], [0.1, 0.005, 0.1]), ([0.625, 0.41, 0.0], [0.1, 0.01, 0.1]), ([0.525, 0.32, 0.0], [0.005, 0.1, 0.1]),
             ([0.725, 0.32, 0.0], [0.005, 0.1, 0.1]), ([0.625, 0.22, 0.0], [0.1, 0.005, 0.1])]
    for i, (pos, size) in enumerate(walls):
        objs.append(prim_body("target_box_%d" % (i + 1), "box", pos, [0, 1, 0, 0], size, mass=0.05, static=True))
    platform = ET.Element("body", name="platform", pos="0.5 -0.1 0.0", quat="1 0 0 0")
    ET.SubElement(platform, "geom", pos="0 0 0", size="0.3 0.3 0.1", type="box", mass="10", friction="0.3 0.001 0.0001", priority="1")
    objs.append(platform)
    return objs


def build_sorting(num_boxes=4):
    m = build_scene("panda_rod_invisible.xml", sorting_objects(num_boxes), "sorting")
    n = num_boxes // 2
    tc = dict(
        n_substeps=35, max_steps=700,                        # sorting.py:195, configs/sorting_4_config.yaml:80
        init_end_eff_pos=[0.525, -0.3, 0.25], init_end_eff_quat=[0, 1, 0, 0],   # sorting_objects.py:11
        rod_geom="rod:geom_rb0", tcp_body="tcp_rb0",
        objects=["red_%d" % (i + 1) for i in range(n)] + ["blue_%d" % (i + 1) for i in range(n)],
        num_boxes=num_boxes,
    )
    return to_blob(m, "sorting", tc)


def aligning_objects():
    """aligning_objects.py:13-66 restated as data: the push box (models/mj/common-objects/robot_push_box/robot_push_box.xml: ONE free body with five
    box geoms - a 10 x 10 x 2 cm plate of 1 kg, friction 0.3 with priority 1, and four 1 g walls) and the target (target_box.xml: sites only, no
    geoms, no joint - it only carries the goal pose); both start at pos 0.6 0.15 0, quat 1 0 0 0 (MjXmlLoadable overrides the file's pose)."""
    objs = []
    for fname, name in (("robot_push_box.xml", "aligning_box"), ("target_box.xml", "target_box")):
        root = ET.parse(os.path.join(D3IL, "models/mj/common-objects/robot_push_box", fname)).getroot()
        body = root.find("worldbody").find("body")
        body.set("name", name)
        body.set("pos", "0.6 0.15 0.0")
        body.set("quat", "1 0 0 0")
        for site in body.findall("site"):
            body.remove(site)
        objs.append(body)
    return objs


def build_aligning():
    m = build_scene("panda_rod_invisible.xml", aligning_objects(), "aligning")
    tc = dict(
        n_substeps=35, max_steps=400,                        # aligning.py:134-135
        init_end_eff_pos=[0.525, -0.35, 0.25], init_end_eff_quat=[0, 1, 0, 0],   # aligning_objects.py:13, aligning.py:274-287
        rod_geom="rod:geom_rb0", tcp_body="tcp_rb0",
        objects=["aligning_box"], target_body="target_box",
        pos_min_dist=0.018, rot_min_dist=0.048, robot_box_dist=0.051,             # aligning.py:215-218
    )
    return to_blob(m, "aligning", tc)


def inserting_objects():
    """gate_insertion_objects.py:5-288 restated as data, in the order gate_insertion.py:229-255 adds them to the scene: three free 5 cm cubes (push_box1..3,
    50 g), three visual-only target boxes (unnamed in the reference: Box(None, ...); they only carry the goal positions), and the seventeen static
    walls maze_3 .. maze_19 of the three gates (maze_1 / maze_2 are constructed but never added: commented out at gate_insertion.py:236-237).  Four walls
    stand diagonally: their quaternions (0, 0.5, +-1, 0) are not unit length - the model compiler normalises them, as MuJoCo does."""
    objs = [
        prim_body("push_box1", "box", [0.4, -0.3, -0.0072], [0, 1, 0, 0], [0.025, 0.025, 0.025], mass=0.05),
        prim_body("push_box2", "box", [0.55, -0.3, -0.0072], [0, 1, 0, 0], [0.025, 0.025, 0.025], mass=0.05),
        prim_body("push_box3", "box", [0.5, -0.35, -0.0072], [0, 1, 0, 0], [0.025, 0.025, 0.025], mass=0.05),
        prim_body("target_box1", "box", [0.3575, 0.276, 0.0], [0, 1, 0, 0], [0.025, 0.025, 0.02], static=True, visual_only=True),
        prim_body("target_box2", "box", [0.525, 0.4535, 0.0], [0, 1, 0, 0], [0.025, 0.025, 0.02], static=True, visual_only=True),
        prim_body("target_box3", "box", [0.6925, 0.276, 0.0], [0, 1, 0, 0], [0.025, 0.025, 0.02], static=True, visual_only=True),
    ]
    maze = {3: ([0.4, 0.17, 0.0], [0, 0.5, 1, 0], [0.03, 0.01, 0.03]), 4: ([0.65, 0.17, 0.0], [0, 0.5, -1, 0], [0.03, 0.01, 0.03]),
            5: ([0.383, 0.2185, 0.0], [0, 1, 0, 0], [0.01, 0.03, 0.03]), 6: ([0.667, 0.2185, 0.0], [0, 1, 0, 0], [0.01, 0.03, 0.03]),
            7: ([0.3525, 0.2385, 0.0], [0, 1, 0, 0], [0.04, 0.01, 0.03]), 8: ([0.6975, 0.2385, 0.0], [0, 1, 0, 0], [0.04, 0.01, 0.03]),
            9: ([0.32, 0.276, 0.0], [0, 1, 0, 0], [0.01, 0.0475, 0.03]), 10: ([0.73, 0.276, 0.0], [0, 1, 0, 0], [0.01, 0.0475, 0.03]),
            11: ([0.3525, 0.3135, 0.0], [0, 1, 0, 0], [0.04, 0.01, 0.03]), 12: ([0.6975, 0.3135, 0.0], [0, 1, 0, 0], [0.04, 0.01, 0.03]),
            13: ([0.383, 0.3335, 0.0], [0, 1, 0, 0], [0.01, 0.03, 0.03]), 14: ([0.667, 0.3335, 0.0], [0, 1, 0, 0], [0.01, 0.03, 0.03]),
            15: ([0.435, 0.3975, 0.0], [0, 0.5, 1, 0], [0.01, 0.07, 0.03]), 16: ([0.615, 0.3975, 0.0], [0, 0.5, -1, 0], [0.01, 0.07, 0.03]),
            17: ([0.4875, 0.4585, 0.0], [0, 1, 0, 0], [0.01, 0.04, 0.03]), 18: ([0.5625, 0.4585, 0.0], [0, 1, 0, 0], [0.01, 0.04, 0.03]),
            19: ([0.525, 0.491, 0.0], [0, 1, 0, 0], [0.0475, 0.01, 0.03])}
    for i in range(3, 20):
        pos, quat, size = maze[i]
        objs.append(prim_body("maze_%d" % i, "box", pos, quat, size, mass=0.05, static=True))
    return objs


def build_inserting():
    """Gate_Insertion_Env (gate_insertion.py:157-282): the rod robot, Cartesian controller, 35 sub-steps, 2000-step episode cap (the constructor default;
    the reference ships no config / Sim class for this task)."""
    m = build_scene("panda_rod_invisible.xml", inserting_objects(), "inserting")
    tc = dict(
        n_substeps=35, max_steps=2000,                       # gate_insertion.py:157-158
        init_end_eff_pos=[0.525, -0.28, 0.12], init_end_eff_quat=[0, 1, 0, 0],   # gate_insertion_objects.py:5, gate_insertion.py:332-356
        rod_geom="rod:geom_rb0", tcp_body="tcp_rb0",
        objects=["push_box1", "push_box2", "push_box3"],
        target_pos=[[0.3575, 0.276, 0.0], [0.525, 0.4535, 0.0], [0.6925, 0.276, 0.0]],    # gate_insertion_objects.py:17-24
        target_min_dist=0.01,                                                              # gate_insertion.py:276
    )
    return to_blob(m, "inserting", tc)


def load_stl_vertices(path):
    """Unique vertices of a binary STL file (models/mj/robot/assets/*.stl)."""
    import struct
    d = open(path, "rb").read()
    n = struct.unpack("<I", d[80:84])[0]
    if 84 + n * 50 != len(d):
        raise ValueError("%s is not a binary STL" % path)
    a = np.frombuffer(d[84:], dtype=np.dtype([("n", "<3f4"), ("v", "<9f4"), ("a", "<u2")]), count=n)
    return np.unique(a["v"].reshape(-1, 3).astype(np.float64), axis=0)


def mesh_hull(path):
    """Collision geometry of a mesh geom: MuJoCo collides the CONVEX HULL of the mesh vertices (qhull at compile time [ext]).
    Returns the hull vertices (mesh file coordinates) and the centroid of the hull volume, which stands in for the geom centre
    MuJoCo derives from the mesh's inertial frame (the centre only seeds the MPR portal search)."""
    from scipy.spatial import ConvexHull
    v = load_stl_vertices(path)
    h = ConvexHull(v)
    hv = v[h.vertices]
    c0 = hv.mean(0)
    vol, cen = 0.0, np.zeros(3)
    for tri in h.simplices:
        a, b, c = v[tri[0]] - c0, v[tri[1]] - c0, v[tri[2]] - c0
        w = abs(np.dot(a, np.cross(b, c))) / 6.0
        vol += w
        cen += w * (a + b + c) / 4.0
    return hv, c0 + cen / vol


def stacking_objects():
    """stacking_objects.py:11-61 restated as data, in the order get_obj_list returns them (red, green, blue, target)."""
    return [
        prim_body("red_box", "box", [0.5, -0.1, 0.0], [0, 1, 0, 0], [0.03, 0.03, 0.03], mass=0.05),
        prim_body("green_box", "box", [0.5, 0.0, 0.0], [0, 1, 0, 0], [0.03, 0.03, 0.03], mass=0.05),
        prim_body("blue_box", "box", [0.5, 0.0, 0.0], [0, 1, 0, 0], [0.03, 0.05, 0.03], mass=0.05),   # init_pos = box_pos2 (sic, stacking_objects.py:44)
        prim_body("target_box", "box", [0.5, 0.2, 0.0], [0, 1, 0, 0], [0.05, 0.05, 0.04], static=True, visual_only=True),
    ]


def build_stacking():
    """CubeStacking_Env (stacking.py:135-198): robot panda_invisible.xml (no rod; finger geoms of class panda:gripper: condim 4,
    margin 1 mm; finger-tip boxes), three free boxes, joint-space PD controller, 30 sub-steps (stacking.py:138)."""
    m = build_scene("panda_invisible.xml", stacking_objects(), "stacking")
    tc = dict(
        n_substeps=30, max_steps=1000,                       # stacking.py:138, configs/stacking_config.yaml:84
        init_end_eff_pos=[0.525, 0.0, 0.3], init_end_eff_quat=[0, 1, 0, 0],     # stacking_objects.py:11, stacking.py:299-317
        tcp_body="tcp_rb0",
        objects=["red_box", "green_box", "blue_box"],
        target_pos=[0.5, 0.2, 0.0], pos_min_dist=0.06,       # stacking_objects.py:17, stacking.py:193
        gripper_open_threshold=0.075,                        # stacking.py:337
        collision_meshes=["fingerv", "handv"],           # mesh geoms whose hull is carried: finger <-> box grasp contacts, hand (palm) <-> box
    )
    blob = to_blob(m, "stacking", tc)
    blob["meshes"] = {}
    for name in tc["collision_meshes"]:      # panda_invisible.xml:72 (panda_hand:geom2, mesh handv: 773 hull vertices), :99 / :108 (finger geoms, mesh fingerv: 68)
        hv, cen = mesh_hull(os.path.join(D3IL, "models/mj/robot/assets/%s.stl" % name))
        blob["meshes"][name] = dict(vert=hv.tolist(), center=cen.tolist())
    # bounding box of the hand mesh in its geom frame: the engine's exact cull of the box <-> hand pairs (the 773-vertex MPR only runs inside it)
    hand = load_stl_vertices(os.path.join(D3IL, "models/mj/robot/assets/handv.stl"))
    blob["task_const"]["hand_bbox_min"] = hand.min(0).tolist()
    blob["task_const"]["hand_bbox_max"] = hand.max(0).tolist()
    return blob


def main():
    out_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "blobs")
    os.makedirs(out_dir, exist_ok=True)
    blob = build_avoiding()
    with open(os.path.join(out_dir, "avoiding.json"), "w") as f:
        json.dump(blob, f, indent=1)
    print("avoiding: %d bodies, %d geoms, %d actuators" % (len(blob["bodies"]), len(blob["geoms"]), len(blob["actuators"])))
    blob = build_pushing()
    with open(os.path.join(out_dir, "pushing.json"), "w") as f:
        json.dump(blob, f, indent=1)
    print("pushing: %d bodies, %d geoms, %d actuators" % (len(blob["bodies"]), len(blob["geoms"]), len(blob["actuators"])))
    blob = build_sorting(4)
    with open(os.path.join(out_dir, "sorting.json"), "w") as f:
        json.dump(blob, f, indent=1)
    print("sorting-4: %d bodies, %d geoms, %d actuators" % (len(blob["bodies"]), len(blob["geoms"]), len(blob["actuators"])))
    blob = build_sorting(2)
    with open(os.path.join(out_dir, "sorting_2.json"), "w") as f:
        json.dump(blob, f, indent=1)
    print("sorting-2: %d bodies, %d geoms, %d actuators" % (len(blob["bodies"]), len(blob["geoms"]), len(blob["actuators"])))
    blob = build_aligning()
    with open(os.path.join(out_dir, "aligning.json"), "w") as f:
        json.dump(blob, f, indent=1)
    print("aligning: %d bodies, %d geoms, %d actuators" % (len(blob["bodies"]), len(blob["geoms"]), len(blob["actuators"])))
    blob = build_inserting()
    with open(os.path.join(out_dir, "inserting.json"), "w") as f:
        json.dump(blob, f, indent=1)
    print("inserting: %d bodies, %d geoms, %d actuators" % (len(blob["bodies"]), len(blob["geoms"]), len(blob["actuators"])))
    blob = build_stacking()
    with open(os.path.join(out_dir, "stacking.json"), "w") as f:
        json.dump(blob, f, indent=1)
    print("stacking: %d bodies, %d geoms, %d actuators, hull vertices %s" % (len(blob["bodies"]), len(blob["geoms"]), len(blob["actuators"]),
                                                                             {k: len(v["vert"]) for k, v in blob["meshes"].items()}))


if __name__ == "__main__":
    main()
