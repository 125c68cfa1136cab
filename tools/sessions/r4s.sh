#!/bin/bash
# round 4 session s: generic engine with 8 environments per workgroup (twice the workgroups) against 16 - Sorting lines and regimes
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4s; mkdir -p $O
for V in default gl8; do
  if [ $V = default ]; then unset D3IL_LIB_PATH; else export D3IL_LIB_PATH=$PWD/d3il_amd/libd3il_rollout_$V.so; fi
  python tools/gpu_gen_rest_time.py 2>&1 | grep -v amdgpu.ids | sed "s/^/$V: /"
  for SB in 1 4; do
    python bench.py --task sorting --no-cpu-baseline --sub-batches $SB > $O/sorting_${V}_sb$SB.json 2>/dev/null
    python bench.py --task sorting --policy scripted_push --steps 60 --no-cpu-baseline --sub-batches $SB > $O/sorting_scripted_${V}_sb$SB.json 2>/dev/null
  done
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]; print("%-36s %9d env-steps/s  ms %.3f  kernel %.3f  %s" % (f.split("/")[-1][:-5], d["value"], d["ms_per_step"], r["kernel_ms"], d["config"]["flagged_envs"]))
    except Exception as e: print(f, "ERR", e)
PY
