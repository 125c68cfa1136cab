#!/bin/bash
# round 5 session sb: finer sub-batches in the contact regimes (a launch lasts as long as its slowest workgroup; do 8 / 16 / 32 sub-batches hide more of the tail?)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5sb; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
for S in 4 8 16 32; do
  python bench.py --task sorting --policy scripted_push --steps 60 --no-cpu-baseline --sub-batches $S 2>/dev/null | tail -1 > $O/sorting_scripted_sb$S.json
  python bench.py --task pushing --policy scripted_push --steps 100 --no-cpu-baseline --sub-batches $S 2>/dev/null | tail -1 > $O/pushing_scripted_sb$S.json
  python bench.py --task sorting --no-cpu-baseline --sub-batches $S 2>/dev/null | tail -1 > $O/sorting_mlp_sb$S.json
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]; print("%-36s %9d env-steps/s  ms %.3f  kernel %.3f" % (f.split("/")[-1][:-5], d["value"], d["ms_per_step"], r["kernel_ms"]))
    except Exception as e: print(f, "ERR", str(e)[:80])
PY
