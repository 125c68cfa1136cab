"""Build recipe for the in-tree native library (hipcc, gfx950 only).

``libd3il_rollout.so`` is built next to this package so that it travels with the repo snapshot to the
GPU box (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
LIB = os.path.join(PKG, "libd3il_rollout.so")
SOURCES = [os.path.join(PKG, "csrc", "rollout.hip")]
DEPS = SOURCES + [os.path.join(PKG, "csrc", "panda_step.h"), os.path.join(PKG, "csrc", "panda_consts.h"),
                  os.path.join(ROOT, "include", "d3il_rollout.h"), os.path.join(ROOT, "include", "d3il_model_blob.h")]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value"]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP extension cannot be built")
    return exe


def needs_build() -> bool:
    return not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(p) for p in DEPS)


def build(force: bool = False, verbose: bool = False) -> str:
    if force or needs_build():
        cmd = [hipcc()] + HIPCC_FLAGS + ["-o", LIB] + SOURCES
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, cwd=ROOT)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
