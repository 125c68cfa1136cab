"""Instruction-mix summary per kernel of a gfx950 .s file (hipcc -S --cuda-device-only)."""
import collections
import re
import sys

s = open(sys.argv[1]).read().split("\n")
starts = [(i, l.split(":")[0]) for i, l in enumerate(s) if re.match(r"^_Z\w+:", l)]
ends = [i for i, l in enumerate(s) if l.startswith(".Lfunc_end")]
for (i0, name), i1 in zip(starts, ends):
    ops = collections.Counter()
    for l in s[i0:i1]:
        if l.startswith("\t") and not l.startswith("\t.") and not l.startswith("\t;"):
            ops[l.split()[0]] += 1
    tot = sum(ops.values())
    groups = collections.Counter()
    for k, v in ops.items():
        g = ("s_load" if k.startswith("s_load") else "global_load" if k.startswith("global_load") else "global_store" if k.startswith("global_store")
             else "scratch_load" if k.startswith("scratch_load") else "scratch_store" if k.startswith("scratch_store")
             else "v_f64" if k.endswith("_f64") or "_f64_" in k else "v_writelane/readlane" if k in ("v_writelane_b32", "v_readlane_b32")
             else "accvgpr" if "accvgpr" in k else "s_waitcnt" if k == "s_waitcnt" else "s_nop" if k == "s_nop" else "v_mov" if k.startswith("v_mov") else "s_other" if k.startswith("s_") else "v_other")
        groups[g] += v
    print("%s\n  total %d  %s" % (name[:60], tot, dict(groups.most_common())))
