"""Host build of the generic engine against the oracle along the scripted push of tests/test_sorting_host.py: prints the state error
per env step (CPU only).  Usage: python tools/host/sort_host_vs_oracle.py [n_steps]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.oracle import Oracle
from tests.hostcheck.hostcheck import GenHostCheck
from tests.test_sorting_host import _state_err
from tests.test_sorting_oracle import CTX
from d3il_amd.model import blob as blob_mod


def main():
    nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 170
    b = blob_mod.load("sorting")
    q0 = np.load(os.path.join(ROOT, "tests", "golden", "ref_offline_ik.npz"))["sorting__traj_last"].copy()
    o = Oracle(b)
    o.env_start(q0)
    h = GenHostCheck(b)
    obs = o.sort_reset(CTX)
    h.reset(q0, CTX)
    z = float(o.body(b.tcp_body)[0][2])
    des = obs[:2].astype(float)
    worst = 0.0
    for t in range(nsteps):
        box = obs[2:4].astype(float)
        if t < 12:
            target = des.copy()
        else:
            aligned = abs(des[0] - box[0]) < 0.008 and des[1] < box[1] - 0.02
            target = np.array([box[0], 0.36]) if aligned else box + np.array([0.0, -0.06])
        d = target - des
        n = np.linalg.norm(d)
        des = des + d / max(n, 1e-9) * min(0.006, n)
        a = np.concatenate([des, [z], [0, 1, 0, 0]])
        obs, done, info = o.sort_step(a)
        oh, dh, ih = h.step(a)
        e = _state_err(h, o)
        worst = max(worst, e)
        if t % 10 == 0 or info["mode"] != 240:
            print("step %3d  err %.3e  contacts %d  mode %d/%d flags %x" % (t, e, len(o.contacts()), info["mode"], ih["mode"], ih["flags"] & 0x1F0000))
        if info["mode"] != 240:
            break
    print("worst %.3e" % worst)


if __name__ == "__main__":
    main()
