"""Host-side (no GPU) histogram of the joint island solves of the generic engine under the scripted push policy of
bench.py --task sorting --policy scripted_push: which island shapes the contact regime consists of, and their Newton iterations.
Usage: python tools/host/sort_island_hist.py [n_contexts] [n_steps]"""
import ctypes as C
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from d3il_amd.agents import ScriptedPushPolicy
from d3il_amd.envs.sorting import sample_contexts
from d3il_amd.model import blob as blob_mod
from tests.hostcheck.hostcheck import GenHostCheck, lib, _p


def main():
    nctx = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    task = sys.argv[3] if len(sys.argv) > 3 else "sorting"
    b = blob_mod.load(task)
    from d3il_amd.controllers.offline_ik import offline_ik
    from d3il_amd.kinematics import UrdfChain
    js = blob_mod.load_json(task)
    c_, tc_ = js["controller"], js["task_const"]
    q, _, _ = offline_ik(UrdfChain(js["urdf_chain"]), c_["default_qpos"], list(tc_["init_end_eff_pos"]) + list(tc_["init_end_eff_quat"]), np.array(c_["joint_pos_min"]), np.array(c_["joint_pos_max"]))
    if task == "sorting":
        ctx = sample_contexts(60, 4, seed=0)
    else:
        from d3il_amd.envs.inserting import sample_contexts as sic
        ctx = sic(60, seed=0)
    L = lib()
    hist = np.zeros(80, dtype=np.int64)
    L.hc_gen_island_hist(_p(hist), 1)
    total_sub = 0
    for i in range(nctx):
        h = GenHostCheck(b)
        obs = h.reset(q, ctx[i])
        pol = ScriptedPushPolicy(task, device="cpu")
        des = np.array(obs[:2], dtype=float)
        z = None
        # TCP z right after reset: state rows 25..27 hold the TCP
        z = float(h.s[27])
        for t in range(nsteps):
            oin = torch.as_tensor(np.concatenate([des, obs.astype(float)])[None])
            des = des + pol.predict_batch(oin)[0].numpy()
            a = np.concatenate([des, [z], [0, 1, 0, 0]])
            obs, done, info = h.step(a)
            total_sub += 35
            if done:
                break
        fl = info["flags"]
        print("ctx %d: steps %d flags %x mode %d" % (i, t + 1, fl & 0x1F0000, info["mode"]), flush=True)
    L.hc_gen_island_hist(_p(hist), 0)
    print("sub-steps:", total_sub)
    print("joint solves by cube <-> cube pairs in contact inside the island: 0: %d  1: %d  2: %d  3: %d" % tuple(hist[36:40]))
    for key in range(36):
        if hist[key]:
            print("cubes %d arm %d rod-contacts %d: %8d solves (%.3f per sub-step), %.2f Newton iterations each" %
                  (key % 5, (key // 5) % 2, key // 10, hist[key], hist[key] / total_sub, hist[40 + key] / hist[key]))


if __name__ == "__main__":
    main()
