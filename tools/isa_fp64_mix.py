"""FP64 instruction mix of the step kernels, COUNTED in the gfx950 ISA (replaces the 0.66 x 1.6 guess of round 2).

    python tools/isa_fp64_mix.py [out.json]        (needs hipcc; runs where the sources are)

Compiles d3il_amd/csrc/rollout.hip to assembly with the flags of d3il_amd/build.py and counts, per step kernel (its own body plus every device
function of the translation unit that belongs to its task, since the non-inlined solver functions carry most of the arithmetic of Pushing / Sorting):
VALU instructions, FP64 FMA-class instructions (v_fma_f64, v_fmac_f64: 2 flop per lane), other FP64 arithmetic (add, mul, min, max, rcp, rsq, sqrt,
div_*: 1 flop per lane).  `flop_per_valu` = (2 fma + other) / valu is what bench.py multiplies the PMC-counted dynamic VALU instructions with.
It is a STATIC mix: loops and branches weight it differently at run time; the line in bench.py says so."""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from d3il_amd import build  # noqa: E402

TASK_FUNCS = {
    "avoiding": ("k_avoiding_step_split", ["jacobi_solve6", "solve_constraints"]),
    "pushing": ("k_sorting_step", ["gen_", "jacobi_solve6", "solve_constraints"]),      # Pushing runs on the generic engine since round 5
    "sorting": ("k_sorting_step", ["gen_", "jacobi_solve6", "solve_constraints"]),
    "stacking": ("k_stacking_step", []),      # (the solver is inlined since the wrench-form rewrite of round 6)
    "aligning": ("k_aligning_step", ["jacobi_solve6"]),
    "inserting": ("k_sorting_step", ["gen_", "jacobi_solve6", "solve_constraints"]),      # the Sorting kernels run the Inserting model
}
FMA = re.compile(r"^v_(fma|fmac|mad)_f64")
F64 = re.compile(r"^v_\w+_f64")


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r06", "isa_fp64_mix.json")
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "rollout.s")
        flags = [f for f in build.HIPCC_FLAGS if f not in ("-fPIC", "-shared", "-Xarch_device")]
        subprocess.check_call([build.hipcc()] + flags + ["--cuda-device-only", "-S", "-o", asm] + build.SOURCES, cwd=ROOT, stderr=subprocess.DEVNULL)
        text = open(asm).read()
    parts = re.split(r"^\t\.type\t(\w+),@function\n", text, flags=re.M)
    funcs = {}
    for k in range(1, len(parts), 2):
        body = parts[k + 1]
        end = body.find("; -- End function")
        c = dict(valu=0, fma=0, f64_other=0, salu=0, lds=0, vmem=0)
        for line in (body[:end] if end >= 0 else body).split("\n"):
            if not line.startswith("\t") or line.startswith("\t.") or line.startswith("\t;"):
                continue
            op = line.split()[0]
            if op.startswith("v_"):
                c["valu"] += 1
                if FMA.match(op):
                    c["fma"] += 1
                elif F64.match(op) and not op.startswith("v_cmp") and not op.startswith("v_cvt") and not op.startswith("v_mov"):
                    c["f64_other"] += 1
            elif op.startswith("s_"):
                c["salu"] += 1
            elif op.startswith("ds_"):
                c["lds"] += 1
            elif op.startswith(("global_", "scratch_", "flat_", "buffer_")):
                c["vmem"] += 1
        funcs[parts[k]] = c
    res = {}
    for task, (kern, extra) in TASK_FUNCS.items():
        tot = dict(valu=0, fma=0, f64_other=0, salu=0, lds=0, vmem=0)
        used = []
        for name, c in funcs.items():
            if kern in name and ("ILb0E" not in name) or any(e in name for e in extra):
                used.append(name)
                for key in tot:
                    tot[key] += c[key]
        tot["flop_per_valu"] = (2 * tot["fma"] + tot["f64_other"]) / max(1, tot["valu"])
        tot["fp64_share_of_valu"] = (tot["fma"] + tot["f64_other"]) / max(1, tot["valu"])
        tot["functions"] = len(used)
        res[task] = tot
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps({k: {"flop_per_valu": round(v["flop_per_valu"], 3), "fp64_share": round(v["fp64_share_of_valu"], 3), "valu": v["valu"]} for k, v in res.items()}))


if __name__ == "__main__":
    main()
