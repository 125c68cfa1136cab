#!/bin/bash
# round 5 session lw: the two single-node solves compiled with expression-level contraction (gen_tree.h) - is the per-WAVE choice of the lone-cube solver
# now position independent?  Default build (per-environment choice) and the `lonewave` variant side by side.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5lw; mkdir -p $O
T="tests/test_gpu_parity_sorting.py tests/test_gpu_permutation.py tests/test_gpu_parity_pushing.py tests/test_gpu_parity_inserting.py"
timeout 1200 python -m pytest $T -x -q -m gpu > $O/pytest_default.log 2>&1; tail -2 $O/pytest_default.log
export D3IL_LIB_PATH=$PWD/d3il_amd/libd3il_rollout_lonewave.so
timeout 900 python tools/gpu_perm_push_sort.py > $O/perm_lonewave.log 2>&1; grep -i "differ\|identical" $O/perm_lonewave.log | tail -6
timeout 1200 python -m pytest $T -q -m gpu > $O/pytest_lonewave.log 2>&1; tail -4 $O/pytest_lonewave.log
for V in default lonewave; do
  if [ $V = default ]; then unset D3IL_LIB_PATH; else export D3IL_LIB_PATH=$PWD/d3il_amd/libd3il_rollout_lonewave.so; fi
  python bench.py --task sorting --no-cpu-baseline 2>/dev/null | tail -1 > $O/${V}_sorting_mlp_sb4.json
  python bench.py --task sorting --policy scripted_push --steps 60 --no-cpu-baseline --sub-batches 1 2>/dev/null | tail -1 > $O/${V}_sorting_scripted_sb1.json
  python bench.py --task pushing --no-cpu-baseline 2>/dev/null | tail -1 > $O/${V}_pushing_mlp_sb4.json
  python bench.py --task pushing --policy scripted_push --steps 100 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${V}_pushing_scripted_sb4.json
  python bench.py --task inserting --policy scripted_push --steps 60 --warmup 5 --preroll 300 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${V}_inserting_scripted.json
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]; print("%-40s %9d env-steps/s  ms %.3f  kernel %.3f  %s" % (f.split("/")[-1][:-5], d["value"], d["ms_per_step"], r["kernel_ms"], d["config"]["flagged_envs"]))
    except Exception as e: print(f, "ERR", str(e)[:60])
PY
