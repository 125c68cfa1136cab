"""Histogram of a disassembly with line info (llvm-objdump -d -l): instructions per source line inside functions whose name contains argv[2]."""
import collections
import re
import sys

path, want = sys.argv[1], sys.argv[2]
cur, infn = None, False
hist = collections.Counter(); f64 = collections.Counter(); scr = collections.Counter(); tot = 0
for l in open(path):
    m = re.match(r"^[0-9a-f]+ <(.*)>:", l)
    if m:
        infn = want in m.group(1); continue
    if not infn:
        continue
    m = re.match(r"^; (\S+):(\d+)", l)
    if m:
        cur = (m.group(1).split("/")[-1], int(m.group(2))); continue
    if re.match(r"^\s+[a-z]", l):
        hist[cur] += 1; tot += 1
        if "_f64" in l: f64[cur] += 1
        if "scratch_" in l or "accvgpr" in l: scr[cur] += 1
print("total instructions", tot, "f64", sum(f64.values()), "scratch+acc", sum(scr.values()))
for k, v in hist.most_common(45):
    print("%-28s %6d  f64 %6d  spill-ish %5d" % ("%s:%d" % k if k else "?", v, f64[k], scr[k]))
