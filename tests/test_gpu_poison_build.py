"""The NaN-poison build of the cooperative engine (Stacking / Aligning): every LDS word of a workgroup starts as a NaN (and, for Stacking, the dead areas are
poisoned again every sub-step).  A phase that reads a word its launch has not written then produces a NaN on every box - with the product build the same read
returns whatever the LDS held before, which differs from box to box (round 6: the finger-slide axes of the rod-robot variants were read by the wrench-form
solver although stack_pre_kin writes them for the gripper robot only; some boxes of the pool ran the Aligning tests green, others saw NaN states: DESIGN 20.9).
The parity files of both tasks run in a child process with D3IL_LIB_PATH pointing at the poison library (built by __graft_entry__.build())."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
POISON = os.path.join(ROOT, "d3il_amd", "libd3il_rollout_poison.so")


@pytest.mark.gpu
@pytest.mark.parametrize("files", [("tests/test_gpu_parity_aligning.py",), ("tests/test_gpu_parity_stacking.py", "tests/test_gpu_permutation.py")])
def test_parity_files_pass_on_the_poison_build(files):
    if os.environ.get("D3IL_LIB_PATH"):
        pytest.skip("already running on a variant library")
    if not os.path.exists(POISON):
        pytest.skip("libd3il_rollout_poison.so not built (python -c 'from d3il_amd import build; build.build_poison()')")
    lib = os.path.join(ROOT, "d3il_amd", "libd3il_rollout.so")
    assert os.path.getmtime(POISON) >= os.path.getmtime(lib) - 3600, "the poison library is older than the product library: rebuild it"
    env = dict(os.environ, D3IL_LIB_PATH=POISON)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-x", "-p", "no:cacheprovider"] + list(files), cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = "\n".join(r.stdout.splitlines()[-15:])
    assert r.returncode == 0, tail
