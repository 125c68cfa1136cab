"""Binary model blob: the POD struct handed to ``d3il_create`` (include/d3il_rollout.h).

One table (``FIELDS``) defines the layout; ``emit_header()`` generates
``include/d3il_model_blob.h`` from it and ``pack()`` fills a ``ctypes`` mirror of the
same struct from the JSON produced by ``mjcf_compile.py``.  The blob carries *model
data only* (what MuJoCo's MjModel + the gin file + the URDF hold in the reference:
mj_scene_parser.py:36-53, MjFactory.py:24-28, Model.py:26-35); every derived
quantity (fused inertias, invweights, ...) is computed by its consumer.
"""
from __future__ import annotations

import ctypes as C
import json
import os

MAGIC = 0x4C493344  # 'D3IL'
VERSION = 5

MAXBODY, MAXJNT, MAXGEOM, MAXACT, MAXEXCL, MAXCHAIN, MAXOBST = 64, 16, 80, 12, 16, 16, 8
MAXMESH, MAXMVERT = 2, 800

TASK_IDS = {"avoiding": 0, "pushing": 1, "sorting": 2, "stacking": 3, "aligning": 4, "inserting": 5}
JNT_TYPES = {"free": 0, "hinge": 2, "slide": 3}       # numeric values follow mjtJoint [ext]
GEOM_TYPES = {"plane": 0, "sphere": 2, "cylinder": 5, "box": 6, "mesh": 7}  # mjtGeom [ext]

I32, U32, F64 = C.c_int32, C.c_uint32, C.c_double

# (name, ctype, shape, comment)
FIELDS = [
    ("magic", U32, (), "D3IL_BLOB_MAGIC"),
    ("version", U32, (), "D3IL_BLOB_VERSION"),
    ("task_id", I32, (), "0 avoiding, 1 pushing, 2 sorting, 3 stacking, 4 aligning, 5 inserting"),
    ("nbody", I32, (), ""), ("njnt", I32, (), ""), ("ngeom", I32, (), ""),
    ("nu", I32, (), ""), ("nexclude", I32, (), ""), ("nchain", I32, (), ""),
    ("iterations", I32, (), "solver iteration cap (MuJoCo default 100)"),
    ("n_substeps", I32, (), "physics sub-steps per env step"),
    ("max_steps", I32, (), "episode cap (env steps)"),
    ("tcp_body", I32, (), "body whose xpos/xquat is the TCP (MjRobot.py:138)"),
    ("rod_geom", I32, (), "geom id of the rod (mj_helper.py:15-17), -1 if none"),
    ("n_obst", I32, (), ""),
    ("ik_num_iter", I32, (), ""),
    ("obst_geom", I32, (MAXOBST,), "obstacle geom ids, order of avoiding.py:204-218"),
    ("pad0", I32, (2,), "keeps the doubles 8-byte aligned"),
    # options (base.xml:3)
    ("timestep", F64, (), ""), ("gravity", F64, (3,), ""), ("impratio", F64, (), ""),
    ("tolerance", F64, (), ""),
    # bodies
    ("body_parent", I32, (MAXBODY,), ""), ("body_jntadr", I32, (MAXBODY,), "-1 if none"),
    ("body_jntnum", I32, (MAXBODY,), ""), ("body_pad", I32, (MAXBODY,), ""),
    ("body_pos", F64, (MAXBODY, 3), ""), ("body_quat", F64, (MAXBODY, 4), "w x y z, normalised"),
    ("body_mass", F64, (MAXBODY,), ""), ("body_ipos", F64, (MAXBODY, 3), ""),
    ("body_iquat", F64, (MAXBODY, 4), ""), ("body_inertia", F64, (MAXBODY, 3), "diagonal, inertial frame"),
    # joints
    ("jnt_type", I32, (MAXJNT,), "0 free, 2 hinge, 3 slide"), ("jnt_body", I32, (MAXJNT,), ""),
    ("jnt_limited", I32, (MAXJNT,), ""), ("jnt_pad", I32, (MAXJNT,), ""),
    ("jnt_axis", F64, (MAXJNT, 3), ""), ("jnt_pos", F64, (MAXJNT, 3), ""),
    ("jnt_range", F64, (MAXJNT, 2), ""), ("jnt_damping", F64, (MAXJNT,), ""),
    ("jnt_solref", F64, (MAXJNT, 2), ""), ("jnt_solimp", F64, (MAXJNT, 5), ""),
    ("jnt_margin", F64, (MAXJNT,), ""),
    # geoms
    ("geom_type", I32, (MAXGEOM,), "0 plane 2 sphere 5 cylinder 6 box 7 mesh"),
    ("geom_body", I32, (MAXGEOM,), ""), ("geom_contype", I32, (MAXGEOM,), ""),
    ("geom_conaffinity", I32, (MAXGEOM,), ""), ("geom_condim", I32, (MAXGEOM,), ""),
    ("geom_priority", I32, (MAXGEOM,), ""),
    ("geom_size", F64, (MAXGEOM, 3), ""), ("geom_pos", F64, (MAXGEOM, 3), ""),
    ("geom_quat", F64, (MAXGEOM, 4), ""), ("geom_friction", F64, (MAXGEOM, 3), ""),
    ("geom_margin", F64, (MAXGEOM,), ""), ("geom_gap", F64, (MAXGEOM,), ""),
    ("geom_solmix", F64, (MAXGEOM,), ""), ("geom_solref", F64, (MAXGEOM, 2), ""),
    ("geom_solimp", F64, (MAXGEOM, 5), ""),
    # actuators (motors, gear 1)
    ("act_jnt", I32, (MAXACT,), ""), ("act_forcelimited", I32, (MAXACT,), ""),
    ("act_forcerange", F64, (MAXACT, 2), ""),
    # <contact><exclude>
    ("exclude", I32, (MAXEXCL, 2), "body pairs"),
    # URDF chain of the controller's kinematic model (Model.py:26-66)
    ("chain_type", I32, (MAXCHAIN,), "0 fixed, 1 revolute"),
    ("chain_xyz", F64, (MAXCHAIN, 3), ""), ("chain_R", F64, (MAXCHAIN, 9), "row-major Rz(y)Ry(p)Rx(r)"),
    ("chain_axis", F64, (MAXCHAIN, 3), ""),
    # controller gains (mujoco_controller_config.gin:6-37) and limits (Robots.py:57-65)
    ("pd_pgain", F64, (7,), ""), ("pd_dgain", F64, (7,), ""),
    ("ik_pgain_pos", F64, (3,), ""), ("ik_pgain_quat", F64, (3,), ""),
    ("ik_pgain_null", F64, (7,), ""), ("ik_rest", F64, (7,), ""), ("ik_W", F64, (7,), ""),
    ("ik_ddgain", F64, (7,), ""), ("ik_J_reg", F64, (), ""), ("ik_filter", F64, (), ""),
    ("ik_min_sv", F64, (), ""), ("ik_max_sv", F64, (), ""), ("ik_lr", F64, (), ""),
    ("ctrl_qmin", F64, (7,), ""), ("ctrl_qmax", F64, (7,), ""), ("default_qpos", F64, (7,), ""),
    # task constants; Avoiding (avoiding.py:94-107): l1_y l2_y l3_y goal_y l1_x l2_top_x l2_bot_x l3_top_x l3_mid_x l3_bot_x
    #   Pushing (pushing_objects.py:10-15, pushing.py:251): target_1 xyz, target_2 xyz, target_min_dist
    ("task_f", F64, (32,), ""),
    # free-joint task objects in observation order (pushing.py:255-280: push_box, push_box2)
    ("n_obj", I32, (), ""), ("obj_pad", I32, (), ""), ("obj_body", I32, (8,), "body ids"),
    # convex hulls of the mesh geoms that take part in collision (Stacking: fingerv.stl, handv.stl of panda_invisible.xml:69-109); vertices in
    # mesh-file coordinates, i.e. in the frame geom_pos / geom_quat place on the body; centre = centroid of the hull volume
    ("nmesh", I32, (), ""), ("mesh_pad", I32, (), ""), ("mesh_nvert", I32, (MAXMESH,), ""),
    ("geom_mesh", I32, (MAXGEOM,), "mesh id of a mesh geom whose hull is carried, -1 otherwise"),
    ("mesh_center", F64, (MAXMESH, 3), ""), ("mesh_vert", F64, (MAXMESH, MAXMVERT, 3), ""),
]


def _ctype(ct, shape):
    for n in reversed(shape):
        ct = ct * n
    return ct


class ModelBlob(C.Structure):
    _fields_ = [(n, _ctype(ct, sh)) for n, ct, sh, _ in FIELDS]


def emit_header() -> str:
    cn = {I32: "int32_t", U32: "uint32_t", F64: "double"}
    lines = [
        "/* GENERATED by d3il_amd/model/blob.py (emit_header) - do not edit by hand.",
        " * Model blob handed to d3il_create(): plain data restating what the reference keeps in",
        " * MjModel (built at mj_scene_parser.py:36-53), the controller gin file",
        " * (mujoco_controller_config.gin:6-37) and the pinocchio URDF (Model.py:26-35). */",
        "#ifndef D3IL_MODEL_BLOB_H", "#define D3IL_MODEL_BLOB_H", "#include <stdint.h>", "",
        "#define D3IL_BLOB_MAGIC 0x%08Xu" % MAGIC, "#define D3IL_BLOB_VERSION %du" % VERSION,
        "#define D3IL_MAXBODY %d" % MAXBODY, "#define D3IL_MAXJNT %d" % MAXJNT,
        "#define D3IL_MAXGEOM %d" % MAXGEOM, "#define D3IL_MAXACT %d" % MAXACT,
        "#define D3IL_MAXEXCL %d" % MAXEXCL, "#define D3IL_MAXCHAIN %d" % MAXCHAIN,
        "#define D3IL_MAXOBST %d" % MAXOBST, "#define D3IL_MAXMESH %d" % MAXMESH, "#define D3IL_MAXMVERT %d" % MAXMVERT, "",
        "enum { D3IL_JNT_FREE = 0, D3IL_JNT_HINGE = 2, D3IL_JNT_SLIDE = 3 };",
        "enum { D3IL_GEOM_PLANE = 0, D3IL_GEOM_SPHERE = 2, D3IL_GEOM_CYLINDER = 5, D3IL_GEOM_BOX = 6, D3IL_GEOM_MESH = 7 };",
        "enum { D3IL_TASK_AVOIDING = 0, D3IL_TASK_PUSHING = 1, D3IL_TASK_SORTING = 2, D3IL_TASK_STACKING = 3, D3IL_TASK_ALIGNING = 4, D3IL_TASK_INSERTING = 5 };",
        "", "typedef struct d3il_model_blob {",
    ]
    for n, ct, sh, cm in FIELDS:
        dims = "".join("[%d]" % k for k in sh)
        lines.append("  %s %s%s;%s" % (cn[ct], n, dims, ("  /* %s */" % cm) if cm else ""))
    lines += ["} d3il_model_blob;", "", "#endif"]
    return "\n".join(lines) + "\n"


def pack(js: dict) -> ModelBlob:
    b = ModelBlob()
    b.magic, b.version = MAGIC, VERSION
    b.task_id = TASK_IDS[js["task"]]
    bodies, geoms = js["bodies"], js["geoms"]
    assert len(bodies) <= MAXBODY and len(geoms) <= MAXGEOM
    b.nbody, b.ngeom = len(bodies), len(geoms)
    opt = js["option"]
    b.timestep, b.impratio, b.tolerance = opt["timestep"], opt["impratio"], opt["tolerance"]
    b.iterations = opt["iterations"]
    for k in range(3):
        b.gravity[k] = opt["gravity"][k]
    jname = {}
    nj = 0
    for i, bd in enumerate(bodies):
        b.body_parent[i] = bd["parent"]
        b.body_jntadr[i] = nj if bd["joints"] else -1
        b.body_jntnum[i] = len(bd["joints"])
        b.body_mass[i] = bd["mass"]
        for k in range(3):
            b.body_pos[i][k] = bd["pos"][k]
            b.body_ipos[i][k] = bd["ipos"][k]
            b.body_inertia[i][k] = bd["inertia"][k]
        for k in range(4):
            b.body_quat[i][k] = bd["quat"][k]
            b.body_iquat[i][k] = bd["iquat"][k]
        for j in bd["joints"]:
            assert nj < MAXJNT
            jname[j["name"]] = nj
            b.jnt_type[nj] = JNT_TYPES[j["type"]]
            b.jnt_body[nj] = i
            b.jnt_limited[nj] = int(j["limited"])
            b.jnt_damping[nj] = j["damping"]
            b.jnt_margin[nj] = j["margin"]
            for k in range(3):
                b.jnt_axis[nj][k] = j["axis"][k]
                b.jnt_pos[nj][k] = j["pos"][k]
            for k in range(2):
                b.jnt_range[nj][k] = j["range"][k]
                b.jnt_solref[nj][k] = j["solreflimit"][k]
            for k in range(5):
                b.jnt_solimp[nj][k] = j["solimplimit"][k]
            nj += 1
    b.njnt = nj
    gname = {}
    for i, g in enumerate(geoms):
        gname[g["name"]] = i
        b.geom_type[i] = GEOM_TYPES[g["type"]]
        b.geom_body[i] = g["body"]
        b.geom_contype[i], b.geom_conaffinity[i] = g["contype"], g["conaffinity"]
        b.geom_condim[i], b.geom_priority[i] = g["condim"], g["priority"]
        b.geom_margin[i], b.geom_gap[i], b.geom_solmix[i] = g["margin"], g["gap"], g["solmix"]
        for k, v in enumerate(g["size"][:3]):
            b.geom_size[i][k] = v
        for k in range(3):
            b.geom_pos[i][k] = g["pos"][k]
            b.geom_friction[i][k] = g["friction"][k]
        for k in range(4):
            b.geom_quat[i][k] = g["quat"][k]
        for k in range(2):
            b.geom_solref[i][k] = g["solref"][k]
        for k in range(5):
            b.geom_solimp[i][k] = g["solimp"][k]
    acts = js["actuators"]
    b.nu = len(acts)
    for i, a in enumerate(acts):
        b.act_jnt[i] = jname[a["joint"]]
        b.act_forcelimited[i] = int(a["forcelimited"])
        b.act_forcerange[i][0], b.act_forcerange[i][1] = a["forcerange"]
    b.nexclude = len(js["excludes"])
    for i, (x, y) in enumerate(js["excludes"]):
        b.exclude[i][0], b.exclude[i][1] = x, y
    chain = js["urdf_chain"]
    b.nchain = len(chain)
    for i, c in enumerate(chain):
        b.chain_type[i] = 1 if c["type"] == "revolute" else 0
        for k in range(3):
            b.chain_xyz[i][k] = c["xyz"][k]
            b.chain_axis[i][k] = c["axis"][k]
        for r in range(3):
            for k in range(3):
                b.chain_R[i][3 * r + k] = c["R"][r][k]
    ct = js["controller"]
    cq = ct["cart_pos_quat"]
    for k in range(7):
        b.pd_pgain[k], b.pd_dgain[k] = ct["joint_pd"]["pgain"][k], ct["joint_pd"]["dgain"][k]
        b.ik_pgain_null[k], b.ik_rest[k] = cq["pgain_null"][k], cq["rest_posture"][k]
        b.ik_W[k], b.ik_ddgain[k] = cq["W"][k], cq["ddgain"][k]
        b.ctrl_qmin[k], b.ctrl_qmax[k] = ct["joint_pos_min"][k], ct["joint_pos_max"][k]
        b.default_qpos[k] = ct["default_qpos"][k]
    for k in range(3):
        b.ik_pgain_pos[k], b.ik_pgain_quat[k] = cq["pgain_pos"][k], cq["pgain_quat"][k]
    b.ik_J_reg, b.ik_filter = cq["J_reg"], cq["joint_filter_coefficient"]
    b.ik_min_sv, b.ik_max_sv, b.ik_lr = cq["min_svd_values"], cq["max_svd_values"], cq["learningRate"]
    b.ik_num_iter = cq["num_iter"]
    tc = js["task_const"]
    b.n_substeps, b.max_steps = tc["n_substeps"], tc["max_steps"]
    bname = {bd["name"]: i for i, bd in enumerate(bodies)}
    b.tcp_body = bname[tc["tcp_body"]]
    b.rod_geom = gname[tc["rod_geom"]] if tc.get("rod_geom") else -1
    obst = tc.get("obstacles", [])
    b.n_obst = len(obst)
    for i, o in enumerate(obst):
        b.obst_geom[i] = gname[o + ":geom"]
    if js["task"] == "avoiding":
        keys = ["l1_ypos", "l2_ypos", "l3_ypos", "goal_ypos", "l1_xpos", "l2_top_xpos",
                "l2_bottom_xpos", "l3_top_xpos", "l3_mid_xpos", "l3_bottom_xpos"]
        for i, k in enumerate(keys):
            b.task_f[i] = tc[k]
    objs = tc.get("objects", [])
    b.n_obj = len(objs)
    for i, o in enumerate(objs):
        b.obj_body[i] = bname[o]
    meshes = js.get("meshes", {})
    mesh_id = {}
    for name in tc.get("collision_meshes", []):
        mv = meshes[name]
        assert len(mesh_id) < MAXMESH and len(mv["vert"]) <= MAXMVERT
        k = mesh_id[name] = len(mesh_id)
        b.mesh_nvert[k] = len(mv["vert"])
        for i, v in enumerate(mv["vert"]):
            for c in range(3):
                b.mesh_vert[k][i][c] = v[c]
        for c in range(3):
            b.mesh_center[k][c] = mv["center"][c]
    b.nmesh = len(mesh_id)
    for i, g in enumerate(geoms):
        b.geom_mesh[i] = mesh_id.get(g.get("mesh"), -1) if g["type"] == "mesh" else -1
    if js["task"] == "stacking":
        # stacking_objects.py:17 target position, stacking.py:193 pos_min_dist, stacking.py:337 gripper threshold
        for k in range(3):
            b.task_f[k] = tc["target_pos"][k]
        b.task_f[3], b.task_f[4] = tc["pos_min_dist"], tc["gripper_open_threshold"]
        for k in range(3):
            b.task_f[5 + k], b.task_f[8 + k] = tc["hand_bbox_min"][k], tc["hand_bbox_max"][k]
    if js["task"] == "sorting":
        b.task_f[0] = tc["num_boxes"]
    if js["task"] == "pushing":
        for k in range(3):
            b.task_f[k], b.task_f[3 + k] = tc["target_pos1"][k], tc["target_pos2"][k]
        b.task_f[6] = tc["target_min_dist"]
    if js["task"] == "inserting":
        # gate_insertion_objects.py:17-24 the three target positions (red, green, blue), gate_insertion.py:276 target_min_dist
        for i in range(3):
            for k in range(3):
                b.task_f[3 * i + k] = tc["target_pos"][i][k]
        b.task_f[9] = tc["target_min_dist"]
    if js["task"] == "aligning":
        b.task_f[0], b.task_f[1], b.task_f[2] = tc["pos_min_dist"], tc["rot_min_dist"], tc["robot_box_dist"]
        b.task_f[3] = bname[tc["target_body"]]
    return b


_BLOB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "blobs")


def load_json(task: str) -> dict:
    with open(os.path.join(_BLOB_DIR, task + ".json")) as f:
        return json.load(f)


def load(task: str) -> ModelBlob:
    return pack(load_json(task))


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    with open(os.path.join(root, "include", "d3il_model_blob.h"), "w") as f:
        f.write(emit_header())
    print("sizeof(d3il_model_blob) =", C.sizeof(ModelBlob))
