"""Model blob layout, converter output and the C-ABI surface (no compute calls: runs without a GPU)."""
import ctypes as C
import os
import re

import pytest

from d3il_amd.model import blob

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_is_generated_from_table():
    with open(os.path.join(ROOT, "include", "d3il_model_blob.h")) as f:
        assert f.read() == blob.emit_header()


def test_blob_contents(avoiding_blob, avoiding_json):
    b = avoiding_blob
    assert (b.nbody, b.njnt, b.nu, b.n_obst) == (48, 9, 9, 6)
    assert b.n_substeps == 35 and b.max_steps == 250 and b.timestep == 0.001
    # rod cylinder r = 0.01, half-length 0.15, density 1000 (panda_rod_invisible.xml:77-78)
    rod = [bd for bd in avoiding_json["bodies"] if bd["name"] == "rod_rb0"][0]
    assert rod["mass"] == pytest.approx(3.141592653589793 * 1e-4 * 0.3 * 1000, rel=1e-12)
    # body quats are normalised at compile time
    for bd in avoiding_json["bodies"]:
        assert sum(x * x for x in bd["quat"]) == pytest.approx(1.0, abs=1e-15)
    # gripper class defaults reach the fingertip geoms through childclass
    tips = [g for g in avoiding_json["geoms"] if g["name"].endswith("tip_collision")]
    assert len(tips) == 2 and all(g["margin"] == 0.001 and g["condim"] == 4 and g["solref"] == [0.01, 0.5] for g in tips)


def test_oracle_blob_sizeof_matches(avoiding_blob):
    from oracle.oracle import Oracle
    o = Oracle(avoiding_blob)  # orc_create checks magic/version against the same header
    assert o.nq == 9 and o.nv == 9


def test_capi_library_loads_and_exports_every_declared_symbol():
    from d3il_amd import build, capi
    if not os.path.exists(capi.lib_path()):
        build.build()
    lib = C.CDLL(capi.lib_path())
    with open(os.path.join(ROOT, "include", "d3il_rollout.h")) as f:
        declared = set(re.findall(r"\b(d3il_[a-z_0-9]+)\s*\(", f.read()))
    assert declared == set(capi.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    lib.d3il_blob_sizeof.restype = C.c_size_t
    assert lib.d3il_blob_sizeof() == C.sizeof(blob.ModelBlob)


def test_product_fails_loudly_without_device():
    """No silent CPU fallback: on a machine without a HIP device d3il_create must fail with ENODEVICE."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    from d3il_amd import capi
    L = capi.load()
    b = blob.load("avoiding")
    h = C.c_void_p()
    rc = L.d3il_create(0, 64, 0, C.byref(b), C.sizeof(b), C.byref(h))
    assert rc == -3 and b"no HIP device" in L.d3il_last_error()
    from d3il_amd.envs.avoiding import ObstacleAvoidanceVecEnv
    with pytest.raises(capi.D3ilError):
        ObstacleAvoidanceVecEnv(4, device="cpu")


def _header_enums():
    """{name: value} of every `NAME = value` inside the enums of include/d3il_rollout.h and d3il_model_blob.h (plain C constant expressions)."""
    vals = {}
    for fn in ("d3il_model_blob.h", "d3il_rollout.h"):
        with open(os.path.join(ROOT, "include", fn)) as f:
            text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
        for body in re.findall(r"enum\s*\{(.*?)\}", text, flags=re.S):
            for name, expr in re.findall(r"\b(D3IL_[A-Z0-9_]+)\s*=\s*([^,}]+)", body):
                vals[name] = int(eval(expr.strip(), {"__builtins__": {}}, dict(vals)))
    return vals


def test_python_binding_constants_agree_with_the_header():
    """capi.py mirrors include/d3il_rollout.h by hand (the header had no Stacking layout): every constant the binding
    defines must equal the header's."""
    from d3il_amd import capi
    h = _header_enums()
    pairs = {"STATE_F64": "D3IL_STATE_F64", "STATE_QPOS": "D3IL_STATE_QPOS", "STATE_QVEL": "D3IL_STATE_QVEL", "STATE_BIAS": "D3IL_STATE_BIAS",
             "STATE_TCP": "D3IL_STATE_TCP", "STATE_IK_Q": "D3IL_STATE_IK_Q", "STATE_IK_QD": "D3IL_STATE_IK_QD",
             "FLAG_MODE_MASK": "D3IL_FLAG_MODE_MASK", "FLAG_TERMINATED": "D3IL_FLAG_TERMINATED", "FLAG_SUCCESS": "D3IL_FLAG_SUCCESS",
             "FLAG_ROD_CONTACT": "D3IL_FLAG_ROD_CONTACT", "FLAG_IK_VALID": "D3IL_FLAG_IK_VALID", "FLAG_SOLVER_FAIL": "D3IL_FLAG_SOLVER_FAIL",
             "FLAG_MULTI_CONTACT": "D3IL_FLAG_MULTI_CONTACT",
             "PUSH_STATE_BOX": "D3IL_PUSH_STATE_BOX", "PUSH_STATE_WARM": "D3IL_PUSH_STATE_WARM", "PUSH_STATE_TASK": "D3IL_PUSH_STATE_TASK", "PUSH_STATE_F64": "D3IL_PUSH_STATE_F64",
             "PFLAG_FIRST_MASK": "D3IL_PFLAG_FIRST_MASK", "PFLAG_MODE_MASK": "D3IL_PFLAG_MODE_MASK", "PFLAG_WARM_VALID": "D3IL_PFLAG_WARM_VALID",
             "PFLAG_CON_OVERFLOW": "D3IL_PFLAG_CON_OVERFLOW", "PFLAG_OFF_TABLE": "D3IL_PFLAG_OFF_TABLE",
             "SORT_STATE_BOX": "D3IL_SORT_STATE_BOX", "SORT_STATE_WARM": "D3IL_SORT_STATE_WARM", "SORT_STATE_TASK": "D3IL_SORT_STATE_TASK",
             "SORT_STATE_F64": "D3IL_SORT_STATE_F64",
             "STACK_STATE_BOX": "D3IL_STACK_STATE_BOX", "STACK_STATE_WARM": "D3IL_STACK_STATE_WARM", "STACK_STATE_F64": "D3IL_STACK_STATE_F64",
             "SFLAG_MODE_MASK": "D3IL_SFLAG_MODE_MASK", "SFLAG_WARM_VALID": "D3IL_SFLAG_WARM_VALID", "SFLAG_HAND_NEAR": "D3IL_SFLAG_HAND_NEAR",
             "TASK_AVOIDING": "D3IL_TASK_AVOIDING", "TASK_PUSHING": "D3IL_TASK_PUSHING", "TASK_SORTING": "D3IL_TASK_SORTING",
             "TASK_STACKING": "D3IL_TASK_STACKING", "TALLY_ROW": "D3IL_TALLY_ROW", "TALLY_ALL": "D3IL_TALLY_ALL", "ERCCL": "D3IL_ERCCL"}
    for py, c in pairs.items():
        assert c in h, c
        assert getattr(capi, py) == h[c], (py, getattr(capi, py), h[c])
    assert h["D3IL_STACK_STATE_F64"] == 28 + 3 * 13 + 27 and h["D3IL_SORT_STATE_F64"] == 42 + 4 * 13 + 33 + 2
    # Buffers mirrors d3il_buffers field by field
    with open(os.path.join(ROOT, "include", "d3il_rollout.h")) as f:
        body = re.search(r"typedef struct d3il_buffers \{(.*?)\} d3il_buffers;", re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S), flags=re.S).group(1)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if decl:
            fields += [n.strip().lstrip("*") for n in decl.split(None, 1)[1].split(",")]
    assert fields == [f[0] for f in capi.Buffers._fields_]


def test_build_refuses_an_unvalidated_compiler(monkeypatch):
    """build.check_compiler raises for a hipcc whose HIP version is not the validated one unless D3IL_ALLOW_UNVALIDATED=1 (then it warns)."""
    import subprocess
    import warnings
    from d3il_amd import build as b

    class R:
        stdout = "HIP version: 9.9.12345-abc\nclang version 99\n"
    monkeypatch.setattr(b, "hipcc", lambda: "hipcc")
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: R())
    monkeypatch.delenv("D3IL_ALLOW_UNVALIDATED", raising=False)
    with pytest.raises(RuntimeError, match="validated"):
        b.check_compiler()
    monkeypatch.setenv("D3IL_ALLOW_UNVALIDATED", "1")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert b.check_compiler() == "9.9.12345-abc" and len(w) == 1
    R.stdout = "HIP version: %s.26015-fc0010cf6a\n" % b.VALIDATED_HIP
    monkeypatch.delenv("D3IL_ALLOW_UNVALIDATED", raising=False)
    assert b.check_compiler().startswith(b.VALIDATED_HIP)


def test_tree_solver_header_keeps_expression_level_contraction():
    """gen_tree.h: the lone-cube solver and the tree solver's one-node path must compile to the same arithmetic (the per-wave choice between them,
    gen_kernels.h).  That rests on the header being compiled under `fp contract(on)` and handing the translation unit back under
    `contract(fast)`; the GPU guards are tests/test_gpu_permutation.py and the lane-position test of tests/test_gpu_parity_sorting.py."""
    import os
    import re
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "d3il_amd", "csrc", "gen_tree.h")).read()
    pragmas = re.findall(r"#pragma clang fp contract\((\w+)\)", src)
    assert pragmas == ["on", "fast"]
    on, fast = src.index("fp contract(on)"), src.index("fp contract(fast)")
    for name in ("gen_lone_solve", "gen_tree_solve", "gt_cone_eval", "gt_ldl_n", "gt_ldl_solve_n", "gt_rsqrtd"):      # definitions and calls (not the comments)
        at = [m.start() for m in re.finditer(r"\b%s\s*[(<]" % name, src)]
        assert at and all(on < a < fast for a in at), name
    body = re.sub(r"gt_(cone_eval|ldl_n|ldl_solve_n|rsqrtd)", "", re.sub(r"//[^\n]*", "", src[on:fast]))
    assert not re.search(r"\b(cone_eval|ldl_n\s*<\s*6\s*>|ldl_solve_n\s*<\s*6\s*>|rsqrtd)\s*\(", body)      # the shared helpers' contract(fast) originals are not called from here


def test_ddpm_kernel_entry_validates_before_it_launches():
    """d3il_ddpm_mlp_f32 (include/d3il_rollout.h): shape and pointer checks answer without a device - a denoiser that is not the DDPM configs' shape is
    D3IL_EUNSUPPORTED (policies.DDPMPolicy.fused_ok keeps such models on the torch chain), null / misaligned arguments D3IL_EINVAL, zero rows D3IL_OK."""
    import ctypes as C
    from d3il_amd import capi
    L = capi.load()
    p = [C.c_void_p(4096 + 64 * k) for k in range(12)]      # never dereferenced on these paths
    call = lambda ptrs, rows, sd, T, hid, nb: L.d3il_ddpm_mlp_f32(*ptrs, rows, sd, T, hid, nb, None)
    assert call(p, 16, 16, 4, 128, 4) == -5          # hidden 128
    assert call(p, 16, 19, 4, 256, 4) == -5          # 2 + 8 + 19 inputs do not fit 28
    assert call([None] + p[1:], 16, 16, 4, 256, 4) == -1
    assert call(p, 16, 16, 0, 256, 4) == -1 and call(p, -1, 16, 4, 256, 4) == -1
    assert call(p[:3] + [C.c_void_p(4100)] + p[4:], 16, 16, 4, 256, 4) == -1      # w_in not 16-byte aligned
    assert call(p, 0, 16, 4, 256, 4) == 0
    assert b"d3il_ddpm_mlp_f32" in capi.last_error().encode() if hasattr(capi, "last_error") else True


def test_resmlp_kernel_entry_validates_before_it_launches():
    """d3il_resmlp_f32: the same contract as d3il_ddpm_mlp_f32 - unsupported shapes are D3IL_EUNSUPPORTED (policies.FusedResMLP.ok keeps them on torch's layers)."""
    import ctypes as C
    from d3il_amd import capi
    L = capi.load()
    p = [C.c_void_p(4096 + 64 * k) for k in range(8)]
    call = lambda ptrs, rows, i, h, nb, o: L.d3il_resmlp_f32(*ptrs, rows, i, h, nb, o, None)
    assert call(p, 16, 10, 64, 3, 2) == -5 and call(p, 16, 29, 128, 3, 2) == -5 and call(p, 16, 10, 128, 3, 17) == -5
    assert call([None] + p[1:], 16, 10, 128, 3, 2) == -1 and call(p, 16, 10, 128, -1, 2) == -1
    assert call(p, 0, 10, 256, 4, 2) == 0
