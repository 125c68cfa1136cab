"""Golden Avoiding rollouts produced by the CPU oracle (oracle/d3il_oracle.c): actions in, states / obs /
done / mode / success out.  Regenerate with:  python tests/golden/gen_oracle_rollout.py
Used by the -m gpu parity tests (the GPU box needs no reference and no oracle run for this check) and by
the CPU suite as a regression pin of the oracle itself."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from d3il_amd.model import blob  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402


def main():
    b = blob.load("avoiding")
    iq = np.load(os.path.join(HERE, "ref_offline_ik.npz"))["avoiding__traj_last"]
    out = {"init_qpos": iq}
    for name in ("random", "collide", "succeed", "zigzag"):
        rng = np.random.default_rng(11)
        o = Oracle(b)
        o.env_start(iq)
        obs0 = o.env_reset()
        s0, f0 = o.env_state()
        des = s0[25:28].copy()
        acts, states, flags, obs, done, mode, succ = [], [s0], [f0], [obs0], [False], [np.zeros(9)], [False]
        for t in range(250):
            d = {"random": rng.uniform(-0.01, 0.01, 2), "collide": np.array([0.0005, 0.004]), "succeed": np.array([-0.002, 0.004]),
                 "zigzag": np.array([0.006 * np.sign(np.sin(t / 9.0)), 0.0035])}[name]
            des[:2] += d
            a = np.array([des[0], des[1], s0[27], 0, 1, 0, 0])
            ob, dn, md, sc = o.env_step(a)
            s, f = o.env_state()
            acts.append(a); states.append(s); flags.append(f); obs.append(ob); done.append(dn); mode.append(md); succ.append(sc)
            if dn:
                break
        out[name + "__actions"] = np.array(acts)
        out[name + "__states"] = np.array(states)
        out[name + "__flags"] = np.array(flags)
        out[name + "__obs"] = np.array(obs)
        out[name + "__done"] = np.array(done)
        out[name + "__mode"] = np.array(mode)
        out[name + "__succ"] = np.array(succ)
        print(name, len(acts), "steps, done", done[-1], "success", succ[-1], "mode", mode[-1].astype(int))
    np.savez_compressed(os.path.join(HERE, "oracle_avoiding_rollout.npz"), **out)


if __name__ == "__main__":
    main()
