"""Audit of SGPR-spill VGPRs in a kernel's ISA.

hipcc spills SGPRs into LANES of reserved VGPRs (v_writelane_b32 vS, sX, lane / v_readlane_b32 sX, vS, lane).  Such a VGPR holds 64
unrelated scalars, so any ordinary (EXEC-masked, per-lane) write to it while spill slots are live corrupts scalars, and any
per-lane value kept in it is corrupted in exactly the lanes that are spill slots.  This script lists, for one function of a
`hipcc -S --cuda-device-only` listing,
  * the spill VGPRs (destinations of v_writelane_b32) and how many slots / writes / reads each has,
  * every OTHER instruction that writes one of them, with whether it sits inside a whole-wave region (s_or_saveexec_b64 .., -1),
  * every other instruction that READS one of them (apart from v_readlane_b32 and whole-wave saves).

    python tools/isa_spill_audit.py /tmp/rollout.s k_stacking_step
"""
import re
import sys


def function_body(path, name):
    lines = open(path).read().split("\n")
    start = end = None
    for i, l in enumerate(lines):
        if start is None and re.match(r"^_Z\w*%s\w*:" % name, l):
            start = i
        elif start is not None and l.startswith("\t.size") or (start is not None and ".Lfunc_end" in l and l.endswith(":")):
            end = i
            break
    return lines[start:end], start


def main():
    path, name = sys.argv[1], sys.argv[2]
    body, base = function_body(path, name)
    ins = re.compile(r"^\t(\w+)\s+(.*?)(?:\s*;.*)?$")
    spill = {}
    for l in body:
        m = ins.match(l)
        if m and m.group(1) == "v_writelane_b32":
            ops = [o.strip() for o in m.group(2).split(",")]
            spill.setdefault(ops[0], dict(slots=set(), writes=0, reads=0))
            spill[ops[0]]["slots"].add(ops[2]); spill[ops[0]]["writes"] += 1
    for l in body:
        m = ins.match(l)
        if m and m.group(1) == "v_readlane_b32":
            ops = [o.strip() for o in m.group(2).split(",")]
            if ops[1] in spill:
                spill[ops[1]]["reads"] += 1
    print("function %s: %d lines; spill VGPRs: %s" % (name, len(body), {k: (len(v["slots"]), v["writes"], v["reads"]) for k, v in sorted(spill.items())}))

    def regs_of(op):
        op = op.strip()
        m = re.match(r"^([va])\[(\d+):(\d+)\]$", op)
        if m:
            return ["%s%d" % (m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)]
        m = re.match(r"^([va]\d+)$", op)
        return [m.group(1)] if m else []

    wwm = False
    other_w, other_r = [], []
    for i, l in enumerate(body):
        m = ins.match(l)
        if not m:
            continue
        op, args = m.group(1), m.group(2)
        ops = [o.strip() for o in re.split(r",(?![^\[]*\])", args)]
        if op.startswith("s_or_saveexec_b64") and ops[-1] == "-1":
            wwm = True
            continue
        if op in ("s_mov_b64",) and ops[0] == "exec":
            wwm = False
            continue
        if op in ("v_writelane_b32", "v_readlane_b32"):
            continue
        stores = op.startswith(("scratch_store", "buffer_store", "global_store", "flat_store", "ds_write", "ds_add", "s_", "v_cmp", "v_accvgpr_write")) and not op.startswith("v_cmpx")
        dst = [] if stores or not ops else regs_of(ops[0])
        src = [r for o in (ops if stores else ops[1:]) for r in regs_of(o.split(" ")[0])]
        if op == "v_accvgpr_write_b32":
            src = regs_of(ops[1])
        for r in dst:
            if r in spill:
                other_w.append((base + i + 1, wwm, l.strip()))
        for r in src:
            if r in spill:
                other_r.append((base + i + 1, wwm, l.strip()))
    print("other writes to spill VGPRs: %d (%d outside whole-wave regions)" % (len(other_w), sum(1 for x in other_w if not x[1])))
    for ln, w, t in other_w:
        if not w:
            print("   W line %d: %s" % (ln, t))
    print("other reads of spill VGPRs: %d (%d outside whole-wave regions)" % (len(other_r), sum(1 for x in other_r if not x[1])))
    for ln, w, t in other_r:
        if not w:
            print("   R line %d: %s" % (ln, t))


if __name__ == "__main__":
    main()
