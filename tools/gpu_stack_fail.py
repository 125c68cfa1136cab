"""Stacking: find environments that raise D3IL_FLAG_SOLVER_FAIL under a random policy (BESO with random weights), save the state and action of the
step that raised it, replay it on the device (deterministic?) and on the CPU oracle (what does the reference restatement do from that state?).
usage (GPU box): python tools/gpu_stack_fail.py [steps] [envs]"""
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from d3il_amd import capi  # noqa: E402
from d3il_amd.envs.stacking import CubeStackingVecEnv, load_test_contexts  # noqa: E402
from d3il_amd.model import blob as blob_mod  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 250
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dev = torch.device("cuda:0")
env = CubeStackingVecEnv(n, device=0)
q0, _, _ = env.start()
ctx = load_test_contexts()[:16]
ids = np.arange(n) % 16
env.reset(context=ctx[ids])
pol = bench._random_beso(dev)
last_cmd = env.robot_state().to(torch.float32).clone()
found = []
for t in range(steps):
    st0, fl0, sc0 = env.get_state()
    obs20 = torch.cat((last_cmd, env.obs), dim=1)
    out = pol.predict_batch(obs20).to(torch.float32)
    last_cmd = torch.cat((out[:, :7] + obs20[:, :7], out[:, 7:8]), dim=1)
    act = last_cmd.to(torch.float64).contiguous()
    env.step(act)
    torch.cuda.synchronize()
    fl = env.flags[:n].cpu().numpy()
    bad = np.nonzero((fl & capi.FLAG_SOLVER_FAIL) & ~(fl0 & capi.FLAG_SOLVER_FAIL))[0]
    for e in bad[:4]:
        w0 = int(e) & ~3
        found.append(dict(t=t, e=int(e), state=st0[:, e].copy(), flags=int(fl0[e]), step=int(sc0[e]), action=act[e].cpu().numpy().copy(), ctx=int(ids[e]),
                          wg_state=st0[:, w0:w0 + 4].copy(), wg_flags=fl0[w0:w0 + 4].copy(), wg_step=sc0[w0:w0 + 4].copy(), wg_action=act[w0:w0 + 4].cpu().numpy().copy(),
                          wg_ctx=ids[w0:w0 + 4].copy()))
    if len(found) >= 4:
        break
print("environments that raised SOLVER_FAIL:", [(f["t"], f["e"]) for f in found])
if not found:
    sys.exit(0)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez(os.path.join(ROOT, "gpurun_out", "stack_fail_cases.npz"), **{"case%d_%s" % (i, k): np.asarray(v) for i, f in enumerate(found) for k, v in f.items()})
env.close()
b = blob_mod.load("stacking")
for i, f in enumerate(found):
    env1 = CubeStackingVecEnv(4, device=0)
    env1.start()
    env1.reset(context=ctx[[f["ctx"]] * 4])
    for strict in (0, 1):
        env1.set_option("solver_strict", strict)
        st = np.tile(f["state"][:, None], (1, 4)); flv = np.full(4, f["flags"], dtype=np.uint32); scv = np.full(4, f["step"], dtype=np.int32)
        env1.set_state(st, flv, scv)
        a = torch.as_tensor(np.tile(f["action"], (4, 1)), dtype=torch.float64, device=dev).contiguous()
        env1.step(a); torch.cuda.synchronize()
        s1, f1, _ = env1.get_state()
        print("case %d (t %d env %d): device replay strict=%d -> solver_fail %s, lanes identical %s" % (i, f["t"], f["e"], strict, [bool(x & capi.FLAG_SOLVER_FAIL) for x in f1], bool((s1 == s1[:, :1]).all())))
    # the whole workgroup of the failing environment, in place: does the failure depend on the workgroup mates?
    env4 = CubeStackingVecEnv(4, device=0)
    env4.start(); env4.reset(context=ctx[f["wg_ctx"]])
    outs = []
    for rep in range(3):
        env4.set_state(f["wg_state"], f["wg_flags"].astype(np.uint32), f["wg_step"].astype(np.int32))
        a4 = torch.as_tensor(f["wg_action"], dtype=torch.float64, device=dev).contiguous()
        env4.step(a4); torch.cuda.synchronize()
        s4, f4, _ = env4.get_state()
        outs.append((s4.copy(), f4.copy()))
    print("   workgroup replay x3: solver_fail per env %s ; runs bit-identical %s ; contacts-heavy? finger widths %s" % (
        [[bool(x & capi.FLAG_SOLVER_FAIL) for x in o_[1]] for o_ in outs], all(np.array_equal(outs[0][0], o_[0]) for o_ in outs[1:]), np.round(f["wg_state"][7] + f["wg_state"][8], 4)))
    env4.close()
    if os.environ.get("D3IL_STATS_LIB") == "1":
        import ctypes as C
        recs = []
        envd = CubeStackingVecEnv(4, device=0, n_substeps=1)      # ONE sub-step from the saved state: where do the four copies first differ?
        envd.start(); envd.reset(context=ctx[[f["ctx"]] * 4])
        envd.set_state(st, flv, scv); envd.step(a); torch.cuda.synchronize()
        sd, _, _ = envd.get_state()
        print("      after ONE sub-step: |env0-env1| %.2e |env1-env3| %.2e rows that differ (env1 vs env3): %s" % (np.abs(sd[:, 0] - sd[:, 1]).max(), np.abs(sd[:, 1] - sd[:, 3]).max(), np.nonzero(sd[:, 1] != sd[:, 3])[0][:12]))
        # experiment: only environments 2 and 3 carry the finger-finger jobs (environments 0, 1 get the reset state: open gripper)
        sr, fr, cr = envd.get_state()
        envd.reset(context=ctx[[f["ctx"]] * 4]); torch.cuda.synchronize()
        s0, f0, c0 = envd.get_state()
        stm = st.copy(); stm[:, 0] = s0[:, 0]; stm[:, 1] = s0[:, 1]
        flm = flv.copy(); flm[:2] = f0[:2]; scm = scv.copy(); scm[:2] = c0[:2]
        envd.set_state(stm, flm, scm); envd.step(a); torch.cuda.synchronize()
        sm_, _, _ = envd.get_state()
        print("      jobs only for env 2, 3 (MPR groups 0, 1): |env2-env3| %.2e ; env2 equals the all-four run's env 2: %s" % (np.abs(sm_[:, 2] - sm_[:, 3]).max(), np.array_equal(sm_[:, 2], sd[:, 2])))
        envd.set_state(st, flv, scv); envd.step(a); torch.cuda.synchronize()
        env1_keep = env1
        env1 = envd
        for e_ in range(4):
            buf = np.zeros(620); capi.check(env1.L.d3il_debug_scratch(env1.h, e_, buf.ctypes.data_as(C.c_void_p), len(buf)))
            recs.append(buf)
        for e_ in range(4):
            nc = int(recs[e_][0])
            print("      env %d: ncon %d need %s jsz %s metas %s" % (e_, nc, recs[e_][1], recs[e_][2], [int(recs[e_][8 + 8 * c_ + 7]) for c_ in range(nc)]))

        print("      finger tables env1 vs env3: max diff %.3e ; joint axes / origins: %.3e ; env0 vs env1 tables %.3e" % (np.abs(recs[1][500:548] - recs[3][500:548]).max(), np.abs(recs[1][560:602] - recs[3][560:602]).max(), np.abs(recs[0][500:548] - recs[1][500:548]).max()))
        nc = int(recs[1][0])
        if int(recs[3][0]) == nc:
            print("      records env1 vs env3 max diff %.3e" % np.abs(recs[1][8:8 + 8 * nc] - recs[3][8:8 + 8 * nc]).max())
    if os.environ.get("D3IL_STATS_LIB") == "1":
        envd.close(); env1 = env1_keep
    d01 = np.abs(s1[:, 0] - s1[:, 1]).max(); d02 = np.abs(s1[:, 0] - s1[:, 2]).max(); d13 = np.abs(s1[:, 1] - s1[:, 3]).max()
    print("   4 copies in one workgroup: |env0-env1| %.2e |env0-env2| %.2e |env1-env3| %.2e" % (d01, d02, d13))
    o = Oracle(b); o.env_start(q0); o.stack_reset(ctx[f["ctx"]])
    n_mode = f["flags"] & 3
    o.stack_set_state(f["state"][:67], step=f["step"], terminated=bool(f["flags"] & capi.FLAG_TERMINATED), min_inds=[(f["flags"] >> (2 + 2 * k)) & 3 for k in range(n_mode)])
    o.stack_step(f["action"])
    so = o.stack_state()
    cons = o.contacts()
    err = np.abs(s1[:67, 0] - so)
    print("   oracle: contacts %d (deepest %.4f), last solver iterations %d; |device - oracle| max %.3e at row %d; box z %s; arm q %s" % (
        len(cons), min([c[0] for c in cons] + [0]), o.solver_iter(), err.max(), int(err.argmax()), np.round(so[[30, 43, 56]], 4), np.round(so[:9], 3)))
    env1.close()
