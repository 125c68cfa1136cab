"""Turn recorded D3IL demonstrations into golden fixtures for the oracle replay test.

The reference's demonstration pickles (download link in its README; NOT part of its source tree) are the only real-MuJoCo state
traces that exist for this path: every file holds ``env_state['robot']['des_c_pos']`` (the commanded TCP position per env step,
what the rollout loop feeds to env.step) and ``env_state['robot']['c_pos']`` (the TCP position MuJoCo produced), read exactly as
``environments/dataset/avoiding_dataset.py:52-60`` reads them.  This script copies those two arrays (data, not code) of a few
demonstrations into ``tests/golden/demo_<task>.npz``; ``tests/test_demo_replay.py`` then feeds the recorded commands to the oracle
(MjScene.py:110-143 semantics) and compares the TCP trace - the test that can finally pin the mujoco / pinocchio rows (a-3, a-6, a-7).

    python tools/make_demo_fixtures.py /path/to/environments/dataset/data/avoiding/data --task avoiding --n 8
"""
import argparse
import os
import pickle

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("data_dir")
    ap.add_argument("--task", default="avoiding", choices=["avoiding", "pushing", "sorting"])
    ap.add_argument("--n", type=int, default=8, help="number of demonstrations to keep (shortest first: small fixture)")
    args = ap.parse_args()
    files = sorted(f for f in os.listdir(args.data_dir) if not f.startswith("."))
    demos = []
    for f in files:
        with open(os.path.join(args.data_dir, f), "rb") as fh:
            st = pickle.load(fh)
        rob = st["robot"]
        d = dict(name=f, des_c_pos=np.asarray(rob["des_c_pos"], dtype=np.float64), c_pos=np.asarray(rob["c_pos"], dtype=np.float64))
        for k in ("des_c_quat", "c_quat", "j_pos", "j_vel"):            # recorded by RobotLogger when present; optional
            if k in rob:
                d[k] = np.asarray(rob[k], dtype=np.float64)
        for k, v in st.items():                                          # task objects (cube poses): pos / quat traces
            if k != "robot" and isinstance(v, dict) and "pos" in v:
                d["obj_%s_pos" % k] = np.asarray(v["pos"], dtype=np.float64)
                if "quat" in v:
                    d["obj_%s_quat" % k] = np.asarray(v["quat"], dtype=np.float64)
        demos.append(d)
    demos.sort(key=lambda d: len(d["c_pos"]))
    out = {}
    for i, d in enumerate(demos[: args.n]):
        for k, v in d.items():
            if k != "name":
                out["demo%d_%s" % (i, k)] = v
    out["n_demos"] = np.array(min(args.n, len(demos)))
    dst = os.path.join(ROOT, "tests", "golden", "demo_%s.npz" % args.task)
    np.savez_compressed(dst, **out)
    print("wrote %s: %d demonstrations, lengths %s" % (dst, int(out["n_demos"]), [len(d["c_pos"]) for d in demos[: args.n]]))


if __name__ == "__main__":
    main()
