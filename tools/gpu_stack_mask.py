"""Stacking: reset of four copies of one environment under different reset masks (which workgroup positions are live): per live position, does the
state after the reset sub-step equal the reference (position 0 of the all-live reset of the reference build)?
usage (GPU box): D3IL_LIB_PATH=... python tools/gpu_stack_mask.py ref.npy [write]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from d3il_amd.envs.stacking import CubeStackingVecEnv, load_test_contexts  # noqa: E402

ref_path = sys.argv[1]
ctx = load_test_contexts()
n = 8
env = CubeStackingVecEnv(n, device=0)
env.start()
env.reset(context=ctx[[0] * n])
st, _, _ = env.get_state()
if len(sys.argv) > 2:
    np.save(ref_path, st[:, 0])
    print("reference written; all positions identical:", bool((st == st[:, :1]).all()))
    sys.exit(0)
ref = np.load(ref_path)
print("all live: positions equal to the reference:", [bool((st[:, k] == ref).all()) for k in range(n)])
for m in ([0, 0, 0, 1], [0, 0, 1, 1], [0, 1, 0, 1], [1, 0, 0, 1], [1, 1, 1, 0], [0, 0, 1, 0], [0, 1, 1, 1], [1, 1, 0, 1]):
    mask = torch.tensor(m + m, dtype=torch.uint8)
    # disturb the state first so that a skipped reset is visible, then reset the masked positions
    env.reset(context=ctx[[1] * n])
    env.reset(mask=mask, context=ctx[[0] * n])
    st, _, _ = env.get_state()
    print("mask %s: live positions equal to the reference: %s" % (m, {k: bool((st[:, k] == ref).all()) for k in range(n) if (m + m)[k]}))
