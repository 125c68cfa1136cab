#!/bin/bash
# round 5 session k: sub-batches in the product (Sim classes, bench through SubBatchSet), clean phase timers of the sub-lane build
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5k; mkdir -p $O
timeout 1500 python -m pytest tests/test_subbatch_sims.py tests/test_sorting_sim_gpu.py tests/test_pushing_sim_gpu.py tests/test_sims_with_native_policies_gpu.py tests/test_gpu_multiprocess.py -x -q -m gpu > $O/pytest.log 2>&1; tail -5 $O/pytest.log
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/avoiding_sb4.json
python bench.py --no-cpu-baseline --sub-batches 1 2>/dev/null | tail -1 > $O/avoiding_sb1.json
python bench.py --task sorting --no-cpu-baseline 2>/dev/null | tail -1 > $O/sorting_mlp_sb4.json
python bench.py --task pushing --no-cpu-baseline 2>/dev/null | tail -1 > $O/pushing_mlp_sb4.json
python tools/gpu_sort_phases.py 4096 2>&1 | grep -v amdgpu.ids | grep "per workgroup" | tee $O/phases_4096.log
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]; print("%-36s %9d env-steps/s  ms %.3f  kernel %.3f  %s" % (f.split("/")[-1][:-5], d["value"], d["ms_per_step"], r["kernel_ms"], d["config"]["flagged_envs"]))
    except Exception as e: print(f, "ERR", e)
PY
