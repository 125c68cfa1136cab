"""Builds comparison variants of the library in parallel (CPU only): `name=flag,flag,...` pairs, each -> d3il_amd/libd3il_rollout_<name>.so
(git-ignored, travels to the GPU box; selected there with D3IL_LIB_PATH).  Used by the code-generation experiments of DESIGN section 18.2.

    python tools/build_variants.py raw=-DD3IL_SK_PRELOAD_RAW rawnop=-DD3IL_SK_PRELOAD_RAW,-mllvm,-amdgpu-snop-padding=1
"""
import concurrent.futures as cf
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from d3il_amd import build as b  # noqa: E402


def one(spec):
    name, _, flags = spec.partition("=")
    base = list(b.HIPCC_FLAGS)
    extra = [f for f in flags.split(",") if f]
    for f in list(extra):
        if f.startswith("--drop="):          # remove a default flag pair, e.g. --drop=-disable-machine-licm
            extra.remove(f)
            key = f[len("--drop="):]
            i = base.index(key)
            del base[i - 1:i + 1]
    out = os.path.join(b.PKG, "libd3il_rollout_%s.so" % name)
    cmd = [b.hipcc()] + base + extra + ["-o", out] + b.SOURCES
    t0 = time.time()
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True)
    return name, r.returncode, time.time() - t0, (r.stderr or "")[-600:]


if __name__ == "__main__":
    b.generate_consts()
    with cf.ThreadPoolExecutor(max_workers=int(os.environ.get("JOBS", "8"))) as ex:
        for name, rc, dt, err in ex.map(one, sys.argv[1:]):
            print("%-12s rc %d  %.0f s  %s" % (name, rc, dt, err.strip().replace("\n", " | ") if rc else ""))
