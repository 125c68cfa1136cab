#!/bin/bash
# GPU session r4e: the fenced Stacking build - quick permutation probes, long soaks, speed, then the whole GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4e; mkdir -p $O; rm -f $O/summary.log
timeout 300 python tools/gpu_stack_perm.py 2048 30 5 > $O/perm_default.log 2>&1; echo "default rc $? $(tail -1 $O/perm_default.log | cut -c1-200)" >> $O/summary.log
for v in raw nopreload shflreduce; do D3IL_LIB_PATH=$PWD/d3il_amd/libd3il_rollout_$v.so timeout 300 python tools/gpu_stack_perm.py 2048 30 5 > $O/perm_$v.log 2>&1; echo "$v rc $? $(tail -1 $O/perm_$v.log | cut -c1-200)" >> $O/summary.log; done
for seed in 5 7 11; do timeout 900 python tools/gpu_stack_perm.py 8192 300 $seed > $O/soak_default_$seed.log 2>&1; echo "soak default seed $seed rc $? $(tail -1 $O/soak_default_$seed.log | cut -c1-200)" >> $O/summary.log; done
D3IL_LIB_PATH=$PWD/d3il_amd/libd3il_rollout_nopreload.so timeout 900 python tools/gpu_stack_perm.py 8192 300 5 > $O/soak_nopreload.log 2>&1; echo "soak nopreload rc $? $(tail -1 $O/soak_nopreload.log | cut -c1-200)" >> $O/summary.log
timeout 900 python tools/gpu_stack_perm.py 32768 100 3 > $O/soak_default_32768.log 2>&1; echo "soak default 32768 rc $? $(tail -1 $O/soak_default_32768.log | cut -c1-200)" >> $O/summary.log
timeout 900 python bench.py --task stacking --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_stacking.json 2> $O/bench_stacking.err; echo "bench stacking rc $?" >> $O/summary.log
timeout 900 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_avoiding.json 2> $O/bench_avoiding.err; echo "bench avoiding rc $?" >> $O/summary.log
timeout 900 python bench.py --task sorting --steps 60 --warmup 10 --no-cpu-baseline > $O/bench_sorting.json 2> $O/bench_sorting.err; echo "bench sorting rc $?" >> $O/summary.log
timeout 2400 python -m pytest tests/ -x -q -m gpu --deselect tests/test_gpu_count_parity.py::test_sorting_success_and_mode_tables_over_full_episodes > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/summary.log
cat $O/summary.log; tail -15 $O/pytest_gpu.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4e/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config'].get('flagged_envs'))
    except Exception as e: print(f, 'ERR', e)
PY
