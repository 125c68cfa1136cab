#!/bin/bash
# round 5 session u: record blocks with an odd row count (L2 channel / set spread) - WRITE_SIZE / FETCH_SIZE of the Sorting step, bench lines, parity suites; fused rollout tail of Avoiding
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5u; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity_sorting.py tests/test_gpu_parity_inserting.py tests/test_gpu_parity_pushing.py tests/test_gpu_auto_reset.py -x -q -m gpu > $O/pytest.log 2>&1; tail -4 $O/pytest.log
for S in 1 4; do
  for P in mlp scripted_push; do
    X="--task sorting --policy $P --steps 40 --sub-batches $S --no-cpu-baseline"
    timeout 600 rocprofv3 --kernel-include-regex k_sorting_step --pmc WRITE_SIZE --output-format csv -d $O/pw -- python bench.py $X > $O/pmc_w.log 2>&1
    timeout 600 rocprofv3 --kernel-include-regex k_sorting_step --pmc FETCH_SIZE --output-format csv -d $O/pa -- python bench.py $X > $O/pmc_a.log 2>&1
    python tools/pmc_summarize.py k_sorting_step $O/sorting_${P}_sb$S.json $O/pw $O/pa; rm -rf $O/pw $O/pa
  done
done
python bench.py --task sorting --policy scripted_push --steps 60 --no-cpu-baseline --sub-batches 1 2>/dev/null | tail -1 > $O/sorting_scripted_sb1_line.json
python bench.py --task sorting --no-cpu-baseline 2>/dev/null | tail -1 > $O/sorting_mlp_sb4_line.json
python bench.py --task pushing --no-cpu-baseline 2>/dev/null | tail -1 > $O/pushing_mlp_sb4_line.json
for S in 4 8; do for F in 1 0; do GPU_MAX_HW_QUEUES=16 python bench.py --no-cpu-baseline --sub-batches $S --fuse-tail $F 2>/dev/null | tail -1 > $O/avoiding_sb${S}_fuse$F.json; done; done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*line.json")+glob.glob("$O/avoiding*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]; print("%-36s %9d env-steps/s  ms %.3f  kernel %.3f  %s" % (f.split("/")[-1][:-5], d["value"], d["ms_per_step"], r["kernel_ms"], d["config"]["flagged_envs"]))
    except Exception as e: print(f, "ERR", e)
PY
