"""One round of a parallel -opt-bisect-limit search on the RAW Stacking build: builds 7 libraries whose optimisation
pipelines stop after N pass executions, N evenly spaced in (lo, hi), and writes the GPU session script that runs the permutation test on each.

    python tools/bisect_round.py LO HI   ->  d3il_amd/libd3il_rollout_bis<N>.so x 7, tools/sessions/bisect.sh
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lo, hi = int(sys.argv[1]), int(sys.argv[2])
k = min(7, hi - lo - 1)
pts = sorted({lo + (hi - lo) * (i + 1) // (k + 1) for i in range(k)} - {lo, hi})
for f in os.listdir(os.path.join(ROOT, "d3il_amd")):
    if f.startswith("libd3il_rollout_bis"):
        os.remove(os.path.join(ROOT, "d3il_amd", f))
specs = ["bis%d=-DD3IL_SK_PRELOAD_RAW,-mllvm,-opt-bisect-limit=%d" % (n, n) for n in pts]
env = dict(os.environ, JOBS="8")
r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "build_variants.py")] + specs, env=env, capture_output=True, text=True)
print("\n".join(l[:150] for l in r.stdout.splitlines() if l.startswith("bis")))
with open(os.path.join(ROOT, "tools", "sessions", "bisect.sh"), "w") as f:
    f.write("#!/bin/bash\ncd \"$GRAFT_REPO_ROOT\" || exit 1\nexport TMPDIR=/tmp\nO=gpurun_out/bisect; mkdir -p $O; rm -f $O/summary.log\n")
    for n in pts:
        f.write("D3IL_LIB_PATH=$PWD/d3il_amd/libd3il_rollout_bis%d.so timeout 300 python tools/gpu_stack_perm.py 2048 30 5 > $O/perm_%d.log 2>&1; echo \"%d rc $? $(tail -1 $O/perm_%d.log | cut -c1-140)\" >> $O/summary.log\n" % (n, n, n, n))
    f.write("cat $O/summary.log\n")
print("points", pts)
