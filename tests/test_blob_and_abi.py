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
