"""One-step error budget of the generic engine in contact (CPU only).

The host build of the engine (tests/hostcheck: the device's source, lanes run one after the other) and the oracle are advanced ONE physics sub-step from
IDENTICAL states along the scripted push of a Sorting context (the oracle is loaded with the host build's state before every sub-step, so nothing
accumulates).  Per sub-step: the difference of the cubes' positions / velocities after the sub-step under the production stopping rule and under the oracle's
own rule (`solver_strict`), the oracle's optimality residual |M (a - a0) - J' f| at ITS solution, and the number of contacts.  If the difference does not
shrink under the strict rule and is covered by the oracle's own residual divided by the cube's inertia, it is the conditioning of the soft-contact problem in
f64 (stiff rows D ~ 1e6 .. 1e7 next to a cube inertia of 3e-5 kg m^2), not an implementation gap.

    python tools/host/one_step_budget.py [context index] [env steps]      -> table + summary (profiles/r05/one_step_budget.log)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from d3il_amd.agents import ScriptedPushPolicy
from d3il_amd.envs.sorting import sample_contexts
from d3il_amd.model import blob as blob_mod
from oracle.oracle import Oracle
from tests.hostcheck.hostcheck import GenHostCheck, lib

NB = 4
POS = [42 + 13 * b + k for b in range(NB) for k in range(7)]
VEL = [42 + 13 * b + 7 + k for b in range(NB) for k in range(6)]


def run(ci, nsteps, strict):
    """[sub-steps x (|d pos|, |d vel|, oracle residual, contacts)] for context ci under the production (0) / strict (1) stopping rule."""
    js = blob_mod.load_json("sorting")
    js["task_const"]["n_substeps"] = 1
    js["task_const"]["max_steps"] = 10 ** 6
    b1 = blob_mod.pack(js)
    q = np.load(os.path.join(ROOT, "tests", "golden", "ref_offline_ik.npz"))["sorting__traj_last"].copy()
    ctx = sample_contexts(60, 4, seed=0)[ci]
    L = lib()
    L.hc_set_solver_strict(strict)
    try:
        o = Oracle(b1)
        o.env_start(q)
        o.sort_reset(ctx.reshape(4, 7))
        h = GenHostCheck(b1)
        obs = h.reset(q, ctx)
        pol = ScriptedPushPolicy("sorting", device="cpu")
        des, z = np.array(obs[:2], dtype=float), float(h.s[27])
        out = []
        for t in range(35 * nsteps):
            if t % 35 == 0:
                des = des + pol.predict_batch(torch.as_tensor(np.concatenate([des, obs.astype(float)])[None]))[0].numpy()
            a = np.concatenate([des, [z], [0, 1, 0, 0]])
            s0, f0 = h.s.copy(), h.f.copy()
            obs, _, _ = h.step(a)
            o.sort_set_state(s0, int(f0[0]) & 0xFFFFFFFF, int(f0[1]))
            o.sort_step(a)
            qp, qv = o.state()
            dp = max(np.abs(h.box(b)[0] - qp[7 * b:7 * b + 3]).max() for b in range(NB))
            dv = max(np.abs(h.box(b)[2] - qv[6 * b:6 * b + 6]).max() for b in range(NB))
            res = float(np.abs(o.grad_at(o.vec("qacc"))).max())
            out.append((dp, dv, res, len(o.contacts())))
    finally:
        L.hc_set_solver_strict(0)
    return np.array(out)


def main():
    ci = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 90
    rows = {strict: run(ci, nsteps, strict) for strict in (0, 1)}
    prod, strict = rows[0], rows[1]
    print("context %d, %d sub-steps, each from identical states (host build of the generic engine vs oracle)" % (ci, len(prod)))
    for name, r in (("production rule", prod), ("oracle's rule (solver_strict)", strict)):
        print("%-30s |d pos| median %.1e  p99 %.1e  max %.1e     |d vel| median %.1e  p99 %.1e  max %.1e" %
              (name, np.median(r[:, 0]), np.percentile(r[:, 0], 99), r[:, 0].max(), np.median(r[:, 1]), np.percentile(r[:, 1], 99), r[:, 1].max()))
    worst = np.argsort(strict[:, 1])[-10:]
    print("oracle's own optimality residual |M (a - a0) - J' f| at its solution: median %.1e  max %.1e N; in the ten worst sub-steps %.1e .. %.1e" %
          (np.median(strict[:, 2]), strict[:, 2].max(), strict[worst, 2].min(), strict[worst, 2].max()))
    print("ten worst sub-steps under the strict rule: |d vel| %s" % " ".join("%.1e" % x for x in strict[worst, 1]))
    print("  their residual / cube inertia (3e-5 kg m^2) x dt (1e-3 s):   %s" % " ".join("%.1e" % (x / 3e-5 * 1e-3) for x in strict[worst, 2]))
    print("  contacts in those sub-steps: %s" % " ".join("%d" % x for x in strict[worst, 3]))


if __name__ == "__main__":
    main()
