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
                  os.path.join(PKG, "csrc", "rigid_common.h"), os.path.join(PKG, "csrc", "policy_f16x3.h"), os.path.join(PKG, "csrc", "gen_step.h"), os.path.join(PKG, "csrc", "gen_tree.h"), os.path.join(PKG, "csrc", "gen_kernels.h"),
                  os.path.join(PKG, "csrc", "stack_step.h"), os.path.join(PKG, "csrc", "stack_kernels.h"), os.path.join(PKG, "csrc", "align_step.h"), os.path.join(PKG, "model", "blobs", "stacking.json"),
                  os.path.join(PKG, "csrc", "gen_consts.cpp"), os.path.join(PKG, "model", "blobs", "avoiding.json"),
                  os.path.join(ROOT, "include", "d3il_rollout.h"), os.path.join(ROOT, "include", "d3il_model_blob.h")]
# -disable-machine-licm / -disable-machine-sink: with the model constants baked in as literals, MachineLICM hoists
# their materialisation (s_mov pairs) out of the sub-step loop and then spills ~350 SGPRs through VGPR lanes;
# keeping them next to their use costs nothing (they are re-materialisable) and removes the spills.
# -fno-signed-zeros / -ffinite-math-only (device pass only): value-preserving for finite data; they let the compiler
# drop the multiplications by the exact zeros of the baked kinematic constants (x * 0 -> 0, x + 0 -> x): -17 % FP64
# instructions.  The kernels never produce or consume NaN/Inf on purpose (solver failures are flagged, not encoded).
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
               "-mllvm", "-disable-machine-licm", "-mllvm", "-disable-machine-sink",
               "-Xarch_device", "-fno-signed-zeros", "-Xarch_device", "-ffinite-math-only"]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP extension cannot be built")
    return exe


# The Stacking / Aligning engine was validated (permutation soak, DESIGN sections 17.3 / 18.2) with the hipcc of ROCm 7.2: its optimiser needs the
# convergence fences of stack_step.h around cross-lane operations; another compiler has to pass `python tools/gpu_stack_perm.py 8192 300` again.
VALIDATED_HIP = "7.2"


def check_compiler(verbose: bool = False) -> str:
    """HIP version line of the compiler.  A compiler other than the validated one is REFUSED (the Stacking / Aligning kernels' correctness was established
    per compiler: DESIGN sections 17.3 / 18.2) unless D3IL_ALLOW_UNVALIDATED=1 is set - then it is a warning, and `python tools/gpu_stack_perm.py 8192 300`
    has to pass on a GPU before the kernels are trusted."""
    out = subprocess.run([hipcc(), "--version"], capture_output=True, text=True).stdout
    line = next((l for l in out.splitlines() if l.startswith("HIP version")), "HIP version: unknown")
    ver = line.split(":", 1)[1].strip()
    if not ver.startswith(VALIDATED_HIP):
        msg = ("libd3il_rollout is validated with hipcc of HIP %s.x; this is %s - run tools/gpu_stack_perm.py (DESIGN section 18.2) before trusting the "
               "Stacking / Aligning kernels" % (VALIDATED_HIP, ver))
        if os.environ.get("D3IL_ALLOW_UNVALIDATED") != "1":
            raise RuntimeError(msg + "; set D3IL_ALLOW_UNVALIDATED=1 to build anyway")
        import warnings
        warnings.warn(msg)
    elif verbose:
        print(line)
    return ver


def needs_build() -> bool:
    return not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(p) for p in DEPS)


GEN_SRC = os.path.join(PKG, "csrc", "gen_consts.cpp")
GEN_DIR = os.path.join(PKG, "csrc", "gen")


def generate_consts(verbose: bool = False):
    """Build-time specialisation: derive the constant block of every task model with the library's own host code
    and write it as a constexpr initialiser (csrc/gen/<task>_consts.inc, committed so the GPU box needs no generator run)."""
    import tempfile
    from .model import blob as blob_mod
    os.makedirs(GEN_DIR, exist_ok=True)
    with tempfile.TemporaryDirectory() as td:
        exe = os.path.join(td, "gen_consts")
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-o", exe, GEN_SRC], cwd=ROOT)
        for task, sym in (("avoiding", "kAvoidingConsts"), ("stacking", "kStackingConsts")):
            bin_path = os.path.join(td, task + ".bin")
            with open(bin_path, "wb") as f:
                f.write(bytes(blob_mod.load(task)))
            out = subprocess.check_output([exe, bin_path, sym]).decode()
            dst = os.path.join(GEN_DIR, task + "_consts.inc")
            if not os.path.exists(dst) or open(dst).read() != out:
                with open(dst, "w") as f:
                    f.write(out)
                if verbose:
                    print("wrote", dst)


def build(force: bool = False, verbose: bool = False) -> str:
    if force or needs_build():
        check_compiler(verbose)
        generate_consts(verbose)
        cmd = [hipcc()] + HIPCC_FLAGS + ["-o", LIB] + SOURCES
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, cwd=ROOT)
    return LIB


def build_stats_variant(verbose: bool = False, counts: bool = False, per_wave: bool = False) -> str:
    """Diagnostics library with device-side path counters and phase timers (not used by the product or the tests); counts: also the wave-level event
    counters inside the generic engine's contact loops (d3il_debug_wave_counts; their atomics distort the timers)."""
    out = os.path.join(PKG, "libd3il_rollout_stats.so")
    cmd = [hipcc()] + HIPCC_FLAGS + ["-DD3IL_DEVICE_STATS"] + (["-DD3IL_DEVICE_COUNTS"] if counts else []) + (["-DD3IL_STATS_PER_WAVE"] if per_wave else []) + ["-o", out] + SOURCES
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=ROOT)
    return out


def build_variant(name: str, defines, verbose: bool = False) -> str:
    """Comparison / diagnostics builds of the Stacking kernel, loaded with D3IL_LIB_PATH=<file>:
    poison: every LDS word starts as a NaN and dead areas are poisoned again every sub-step (-DD3IL_SK_POISON);
    raw: without the convergence fences (SK_CONVERGE, stack_step.h) - shows the position-dependence defect of DESIGN sections 17.3 / 18.2 (-DD3IL_SK_PRELOAD_RAW);
    nopreload: the table-reading support function (-DD3IL_SK_NO_PRELOAD)."""
    out = os.path.join(PKG, "libd3il_rollout_%s.so" % name)
    cmd = [hipcc()] + HIPCC_FLAGS + ["-D" + d for d in defines] + ["-o", out] + SOURCES
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=ROOT)
    return out


def build_poison(force: bool = False, verbose: bool = False) -> str:
    """The NaN-poison build of the cooperative engine (libd3il_rollout_poison.so), rebuilt when a source is newer: tests/test_gpu_poison_build.py runs the
    Stacking / Aligning parity files on it - a phase that reads an LDS word its launch has not written shows up as a NaN on EVERY box instead of as a result
    that depends on what the LDS held before (round 6: the slide axes of the rod-robot variants)."""
    out = os.path.join(PKG, "libd3il_rollout_poison.so")
    if force or not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(p) for p in DEPS):
        check_compiler(verbose)
        generate_consts(verbose)
        build_variant("poison", VARIANTS["poison"], verbose)
    return out


VARIANTS = {"gtstatic": ["D3IL_GT_STATIC"],
            "gtinline": ["D3IL_GT_INLINE"],      # the tree solver inlined into the step kernel (no callee-saved register traffic)
            "nsub1": ["D3IL_GEN_NSUB=1"],      # the generic engine with one lane per cube and one physics wave (A/B of the sub-lanes)
            "loneenv": ["D3IL_LONE_PER_ENV=1"],      # lone-cube / tree solver chosen per environment instead of per wave (A/B)
            "poison": ["D3IL_SK_POISON"], "raw": ["D3IL_SK_PRELOAD_RAW"], "nopreload": ["D3IL_SK_NO_PRELOAD"], "poisonraw": ["D3IL_SK_POISON", "D3IL_SK_PRELOAD_RAW"]}

if __name__ == "__main__":
    import sys
    print(build(force=True, verbose=True))
    if "--stats" in sys.argv or "--counts" in sys.argv or "--per-wave" in sys.argv:
        print(build_stats_variant(verbose=True, counts="--counts" in sys.argv, per_wave="--per-wave" in sys.argv))
    for name, defs in VARIANTS.items():
        if "--" + name in sys.argv:
            print(build_variant(name, defs, verbose=True))
