"""Stacking: four copies of one environment in ONE workgroup, one policy step (optionally one sub-step): which state rows differ between the
workgroup positions, and by how much?  usage (GPU box): D3IL_LIB_PATH=... python tools/gpu_stack_pos3.py [n_substeps] [ctx] [steps]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from d3il_amd.envs.stacking import CubeStackingVecEnv, load_test_contexts  # noqa: E402

nsub = int(sys.argv[1]) if len(sys.argv) > 1 else 0
c = int(sys.argv[2]) if len(sys.argv) > 2 else 0
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
kw = dict(n_substeps=nsub) if nsub > 0 else {}
env = CubeStackingVecEnv(4, device=0, **kw)
q0, _, _ = env.start()
ctx = load_test_contexts()
env.reset(context=ctx[[c] * 4])
st0, _, _ = env.get_state()
print("after reset: positions identical", bool((st0 == st0[:, :1]).all()))
act = torch.as_tensor(np.tile(np.concatenate([q0 + 0.01, [0.08]]), (4, 1)), dtype=torch.float64, device="cuda:0").contiguous()
for t in range(steps):
    env.step(act)
    torch.cuda.synchronize()
    st, fl, _ = env.get_state()
    rows = np.nonzero((st != st[:, :1]).any(axis=1))[0]
    print("step %d: rows that differ from position 0: %s" % (t, rows.tolist()))
    for r in rows[:40]:
        print("   row %3d: %s" % (r, " ".join("%+.3e" % (st[r, k] - st[r, 0]) for k in range(4))), " value %.6g" % st[r, 0])
    print("   flags", [hex(int(x)) for x in fl])
if os.environ.get("D3IL_DUMP") == "1":      # diagnostics build: LDS words of the last sub-step per position
    import ctypes as C
    from d3il_amd import capi
    L = capi.load()
    cols = []
    for e in range(4):
        buf = np.zeros(1176)
        capi.check(L.d3il_debug_scratch(env.h, e, buf.ctypes.data_as(C.c_void_p), 1176))
        cols.append(buf)
    cols = np.stack(cols, 1)
    names = [(0, 8, "ncon/need/jsz"), (8, 264, "records"), (300, 364, "rounds"), (500, 548, "tip/hull R p"), (560, 602, "joint axes/origins"), (610, 637, "x"), (637, 664, "a0"), (664, 691, "vel"),
             (700, 745, "M"), (750, 777, "lim"), (780, 830, "q bias tcp bq act ncon need")]
    for a, b, nm in names:
        d = np.nonzero((cols[a:b] != cols[a:b, :1]).any(axis=1))[0]
        print("%-28s differing words: %s" % (nm, [(int(i), ["%.6g" % v for v in cols[a + i]]) for i in d[:12]]))
    if os.environ.get("D3IL_DUMP_MPR") == "1":
        for L_ in (0, 8, 16, 24, 25, 31):
            blk = cols[840 + (L_ % 8) * 40: 840 + (L_ % 8) * 40 + 40, L_ // 8].reshape(4, 10)
            print("lane %2d:" % L_)
            for r in blk:
                print("    dir %s v1 %s v2 %s" % (np.array2string(r[0:3], precision=12), np.array2string(r[3:6], precision=12), np.array2string(r[6:9], precision=12)))
