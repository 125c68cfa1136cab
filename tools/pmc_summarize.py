"""Summarise rocprofv3 --pmc passes (csv output) for one kernel: per-dispatch mean / min / max of every counter.

    python tools/pmc_summarize.py [--bimodal] <kernel-substring> <out.json> <dir> [<dir> ...]

--bimodal: the kernel is launched in two roles of very different size (k_stacking_step: env.step() and, every step, the masked reset
of the few finished environments - most workgroups leave at once); per counter only the launches above 0.2 x the mean of the top
quartile count as step launches (`mean_per_dispatch`), the statistics over all launches are kept under `all_dispatches`.

Each <dir> is the -d directory of one `rocprofv3 --pmc ... --output-format csv` run (counters are collected in separate
passes, never together with the trace domains - see the profiling section of MI355X_MICROARCH.md)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

bimodal = "--bimodal" in sys.argv
if bimodal:
    sys.argv.remove("--bimodal")
kernel, out = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(float))       # counter -> dispatch -> value (summed over XCDs / SEs)
for d in sys.argv[3:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if kernel in row.get("Kernel_Name", ""):
                    acc[row["Counter_Name"]][(f, row["Dispatch_Id"])] += float(row["Counter_Value"])
summary = {}
for c, per in sorted(acc.items()):
    v = list(per.values())
    summary[c] = {"mean_per_dispatch": sum(v) / len(v), "min": min(v), "max": max(v), "dispatches": len(v)}
    if bimodal:
        top = sorted(v)[-max(1, len(v) // 4):]
        thr = 0.2 * sum(top) / len(top)
        big = [x for x in v if x >= thr]
        summary[c] = {"mean_per_dispatch": sum(big) / len(big), "min": min(big), "max": max(big), "dispatches": len(big), "threshold": thr, "all_dispatches": summary[c]}
with open(out, "w") as fh:
    json.dump(summary, fh, indent=1)
print(json.dumps({k: round(v["mean_per_dispatch"], 1) for k, v in summary.items()}))
