import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """Without a HIP device the gpu-marked tests are skipped (plain `pytest tests` stays green off the GPU box)."""
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason="needs a HIP device (MI355X)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def avoiding_blob():
    from d3il_amd.model import blob
    return blob.load("avoiding")


@pytest.fixture(scope="session")
def avoiding_json():
    from d3il_amd.model import blob
    return blob.load_json("avoiding")


@pytest.fixture(scope="session")
def init_qpos():
    import numpy as np
    g = np.load(os.path.join(ROOT, "tests", "golden", "ref_offline_ik.npz"))
    return g["avoiding__traj_last"].copy()


@pytest.fixture()
def oracle(avoiding_blob):
    from oracle.oracle import Oracle
    return Oracle(avoiding_blob)


@pytest.fixture(scope="session")
def hostcheck(avoiding_blob):
    from tests.hostcheck.hostcheck import HostCheck
    return HostCheck(avoiding_blob)


@pytest.fixture(scope="session")
def pushing_blob():
    from d3il_amd.model import blob
    return blob.load("pushing")


@pytest.fixture(scope="session")
def pushing_json():
    from d3il_amd.model import blob
    return blob.load_json("pushing")


@pytest.fixture()
def push_oracle(pushing_blob):
    from oracle.oracle import Oracle
    return Oracle(pushing_blob)


@pytest.fixture(scope="session")
def push_contexts():
    """The 60 evaluation contexts of the reference (environments/dataset/data/pushing/test_contexts.pkl), as the
    2 x (x, y, z=0, quat) rows BlockContextManager.set_context writes into qpos (pushing.py:99-113)."""
    import numpy as np
    c = np.load(os.path.join(ROOT, "tests", "golden", "ref_pushing_task.npz"))["test_contexts"]
    out = np.zeros((len(c), 14))
    out[:, 0:2], out[:, 3:7] = c[:, 0:2], c[:, 3:7]
    out[:, 7:9], out[:, 10:14] = c[:, 7:9], c[:, 10:14]
    return out
