"""Stacking diagnostics: replay selected single-sub-step states (tools/stack_probe_in.npz) on the device and dump the result plus the
solver scratch column (contact records, Newton diagnostics) for comparison with the host build.  Run on the GPU box."""
import ctypes as C
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from d3il_amd import capi  # noqa: E402
from d3il_amd.envs.stacking import CubeStackingVecEnv  # noqa: E402

d = np.load(os.path.join(ROOT, "tools", "stack_probe_in.npz"))
m = len(d["s0"])
out = {}
for batch in (1, 24, 64):          # alone, and as lane 5 of a batch whose other lanes hold the other probe states (divergent neighbours)
    env = CubeStackingVecEnv(batch, device=0, n_substeps=1, max_steps_per_episode=1000000)
    env.set_init_qpos(d["q0"])
    env.reset(context=np.tile(np.concatenate([[0.4, -0.2, 0, 1, 0, 0, 0], [0.4, -0.05, 0, 1, 0, 0, 0], [0.57, -0.1, 0, 1, 0, 0, 0]]), (batch, 1)))
    res, scr = [], []
    for i in range(m):
        st, fl, sc = env.get_state()
        for j in range(batch):
            k = (i + j) % m if j != 5 % batch else i
            st[:, j] = d["s0"][k]; fl[j] = d["f0"][k]; sc[j] = d["c0"][k]
        lane = 5 % batch
        acts = np.stack([d["a"][(i + j) % m if j != lane else i] for j in range(batch)])
        env.set_state(st, fl, sc)
        env.step(torch.as_tensor(acts, dtype=torch.float64, device=env.device).contiguous())
        torch.cuda.synchronize()
        st1, fl1, sc1 = env.get_state()
        buf = np.zeros(32 * 36 + 4)
        capi.check(env.L.d3il_debug_scratch(env.h, lane, buf.ctypes.data_as(C.c_void_p), len(buf)))
        res.append(st1[:, lane].copy()); scr.append(buf)
    out["s1_b%d" % batch] = np.array(res); out["scr_b%d" % batch] = np.array(scr)
    env.close()
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "stack_probe_out.npz"), **out)
print("probe done")
