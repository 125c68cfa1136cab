#!/bin/bash
# round 5 session c: tree solver in point form (world axes, 3-vector contact terms) - parity suites, phases, contact-regime and default bench lines
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/${1:-r5c}; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity_sorting.py tests/test_gpu_parity_inserting.py tests/test_sorting_sim_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; tail -5 $O/pytest.log
python tools/gpu_gen_rest_time.py 2>&1 | grep -v amdgpu.ids | tee $O/rest.log
python tools/gpu_sort_phases.py 4096 2>&1 | grep -v amdgpu.ids | tee $O/phases_4096.log
python bench.py --task sorting --policy scripted_push --steps 60 --no-cpu-baseline --sub-batches 1 2>/dev/null | tail -1 > $O/sorting_scripted_sb1.json
python bench.py --task sorting --policy scripted_push --steps 60 --no-cpu-baseline 2>/dev/null | tail -1 > $O/sorting_scripted_sb4.json
python bench.py --task sorting --no-cpu-baseline 2>/dev/null | tail -1 > $O/sorting_mlp_sb4.json
python bench.py --task sorting --no-cpu-baseline --sub-batches 1 2>/dev/null | tail -1 > $O/sorting_mlp_sb1.json
python bench.py --task inserting --policy scripted_push --steps 60 --warmup 5 --preroll 300 --no-cpu-baseline 2>/dev/null | tail -1 > $O/inserting_scripted.json
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]; print("%-36s %9d env-steps/s  ms %.3f  kernel %.3f  %s" % (f.split("/")[-1][:-5], d["value"], d["ms_per_step"], r["kernel_ms"], d["config"]["flagged_envs"]))
    except Exception as e: print(f, "ERR", e)
PY
