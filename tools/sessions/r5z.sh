#!/bin/bash
# round 5 session z: lone-cube solver built into the step kernel - parity suites, resting times, bench lines, write traffic
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5z; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity_sorting.py tests/test_gpu_parity_inserting.py tests/test_gpu_parity_pushing.py tests/test_gpu_permutation.py tests/test_sorting_sim_gpu.py tests/test_gpu_full_episode_flags.py -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python tools/gpu_gen_rest_time.py 2>&1 | grep -v amdgpu.ids | tee $O/rest.log
for P in mlp; do X="--task sorting --policy $P --steps 40 --sub-batches 4 --no-cpu-baseline"
  timeout 600 rocprofv3 --kernel-include-regex k_sorting_step --pmc WRITE_SIZE --output-format csv -d $O/pw -- python bench.py $X > $O/pmc_w.log 2>&1
  python tools/pmc_summarize.py k_sorting_step $O/lone_sorting_${P}_sb4.json $O/pw; rm -rf $O/pw; done
python bench.py --task sorting --no-cpu-baseline 2>/dev/null | tail -1 > $O/sorting_mlp_sb4.json
python bench.py --task sorting --no-cpu-baseline --sub-batches 1 2>/dev/null | tail -1 > $O/sorting_mlp_sb1.json
python bench.py --task sorting --policy scripted_push --steps 60 --no-cpu-baseline --sub-batches 1 2>/dev/null | tail -1 > $O/sorting_scripted_sb1.json
python bench.py --task sorting --policy ddpm --no-cpu-baseline 2>/dev/null | tail -1 > $O/sorting_ddpm.json
python bench.py --task pushing --no-cpu-baseline 2>/dev/null | tail -1 > $O/pushing_mlp_sb4.json
python bench.py --task pushing --policy scripted_push --steps 100 --no-cpu-baseline 2>/dev/null | tail -1 > $O/pushing_scripted_sb4.json
python bench.py --task inserting --policy scripted_push --steps 60 --warmup 5 --preroll 300 --no-cpu-baseline 2>/dev/null | tail -1 > $O/inserting_scripted.json
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]; print("%-36s %9d env-steps/s  ms %.3f  kernel %.3f  %s" % (f.split("/")[-1][:-5], d["value"], d["ms_per_step"], r["kernel_ms"], d["config"]["flagged_envs"]))
    except Exception as e: print(f, "ERR", str(e)[:60])
PY
