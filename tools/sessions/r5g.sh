#!/bin/bash
# round 5 session g: sub-lane build - parity suites, resting times, then what the physics waves wait for (instruction cache / issue counters)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5g; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity_sorting.py tests/test_gpu_parity_inserting.py tests/test_sorting_sim_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python tools/gpu_gen_rest_time.py 2>&1 | grep -v amdgpu.ids | tee $O/rest.log
python bench.py --task sorting --policy scripted_push --steps 60 --no-cpu-baseline --sub-batches 1 2>/dev/null | tail -1 > $O/sorting_scripted_sb1.json
python bench.py --task sorting --no-cpu-baseline --sub-batches 1 2>/dev/null | tail -1 > $O/sorting_mlp_sb1.json
rocprofv3 --list-avail 2>/dev/null | grep -o -i -E "\b(SQC?_[A-Z_0-9]*(ICACHE|IFETCH|WAIT|BUSY|INST_CYCLES|ACTIVE_INST|INSTS_)[A-Z_0-9]*)\b" | sort -u | tr '\n' ' ' > $O/counters_avail.txt; wc -w $O/counters_avail.txt
X="--task sorting --no-cpu-baseline --sub-batches 1 --steps 30 --warmup 5"
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_IFETCH SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/pmc_$i -- python bench.py $X > $O/pmc_$i.log 2>&1 || echo "set $i failed: $set"
done
python tools/pmc_summarize.py k_sorting_step $O/pmc_issue.json $O/pmc_1 $O/pmc_2 $O/pmc_3 $O/pmc_4 $O/pmc_5
rm -rf $O/pmc_[0-9]
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/sorting*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]; print("%-36s %9d env-steps/s  ms %.3f  kernel %.3f  %s" % (f.split("/")[-1][:-5], d["value"], d["ms_per_step"], r["kernel_ms"], d["config"]["flagged_envs"]))
    except Exception as e: print(f, "ERR", e)
PY
