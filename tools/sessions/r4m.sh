#!/bin/bash
# round 4 session m: the generic engine instantiated with / without contacts of the arm block (Inserting / Sorting): parity tests, regimes, Sorting bench lines (regression check)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4m; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_parity_inserting.py tests/test_gpu_parity_sorting.py tests/test_sorting_sim_gpu.py -x -q -m gpu -s > $O/tests.log 2>&1; grep -n "inserting scripted\|inserting one-step\|passed\|failed" $O/tests.log | tail -8
python tools/gpu_gen_rest_time.py 2>&1 | grep -v amdgpu.ids | tee $O/regimes.log
python bench.py --task sorting --no-cpu-baseline > $O/bench_sorting.json 2>/dev/null
python bench.py --task sorting --policy scripted_push --steps 100 --no-cpu-baseline > $O/bench_sorting_scripted.json 2>/dev/null
python bench.py --task sorting --policy ddpm --no-cpu-baseline > $O/bench_sorting_ddpm.json 2>/dev/null
python bench.py --task inserting --steps 100 --warmup 5 --preroll 300 --no-cpu-baseline > $O/bench_inserting.json 2>/dev/null
python bench.py --task inserting --policy scripted_push --steps 100 --warmup 5 --preroll 300 --no-cpu-baseline > $O/bench_inserting_scripted.json 2>/dev/null
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]; print(f.split("/")[-1], round(d["value"]), "env-steps/s, ms", round(d["ms_per_step"],3), "kernel", round(r["kernel_ms"],3), d["config"]["flagged_envs"])
    except Exception as e: print(f, "ERR", e)
PY
