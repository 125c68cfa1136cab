"""Stacking: A/B two builds of the library that must agree BIT FOR BIT (same arithmetic, different code generation), e.g. the default build
against -DD3IL_SK_NO_PRELOAD.  Each build runs the same seeded random policy (BESO with random weights) in its own process; the states after
every `every` steps are hashed and compared.
usage (GPU box): python tools/gpu_stack_ab.py libA.so libB.so [steps] [envs]      (worker: python tools/gpu_stack_ab.py --worker out.npz steps envs)"""
import hashlib
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(out, steps, n):
    import torch
    import bench
    from d3il_amd import capi
    from d3il_amd.envs.stacking import CubeStackingVecEnv, load_test_contexts
    torch.manual_seed(0)
    dev = torch.device("cuda:0")
    env = CubeStackingVecEnv(n, device=0)
    env.start()
    ctx = load_test_contexts()[:16]
    env.reset(context=ctx[np.arange(n) % 16])
    pol = bench._random_beso(dev)
    last_cmd = env.robot_state().to(torch.float32).clone()
    digests, fails, states = [], [], []
    for t in range(steps):
        obs20 = torch.cat((last_cmd, env.obs), dim=1)
        out_ = pol.predict_batch(obs20).to(torch.float32)
        last_cmd = torch.cat((out_[:, :7] + obs20[:, :7], out_[:, 7:8]), dim=1)
        env.step(last_cmd.to(torch.float64).contiguous())
        torch.cuda.synchronize()
        st, fl, _ = env.get_state()
        digests.append(hashlib.sha256(np.ascontiguousarray(st).tobytes()).hexdigest())
        fails.append(int(((fl & capi.FLAG_SOLVER_FAIL) != 0).sum()))
        states.append(st.copy() if t % 10 == 9 or t == steps - 1 else None)
    np.savez(out, digests=np.array(digests), fails=np.array(fails), last=st, **{"s%d" % t: s for t, s in enumerate(states) if s is not None})


if __name__ == "__main__":
    if sys.argv[1] == "--worker":
        worker(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
        sys.exit(0)
    libs = sys.argv[1:3]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    n = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
    outs = []
    for k, lib in enumerate(libs):
        out = "/tmp/stack_ab_%d.npz" % k
        subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", out, str(steps), str(n)], check=True,
                       env=dict(os.environ, D3IL_LIB_PATH=os.path.abspath(lib)), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        outs.append(np.load(out))
    a, b = outs
    same = a["digests"] == b["digests"]
    first = int(np.argmin(same)) if not same.all() else -1
    print("steps %d envs %d: identical state digests %d of %d; first differing step %d; solver_fail A %d B %d" % (
        steps, n, int(same.sum()), steps, first, int(a["fails"][-1]), int(b["fails"][-1])))
    if first >= 0:
        key = "s%d" % min(t for t in range(first, steps) if "s%d" % t in a.files)
        d = np.nonzero((a[key] != b[key]).any(axis=0))[0]
        print("  at %s: %d environments differ, workgroup positions %s, first %s" % (key, d.size, np.bincount(d % 4, minlength=4).tolist(), d[:8].tolist()))
    sys.exit(0 if same.all() else 1)
