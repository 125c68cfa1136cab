#!/bin/bash
# GPU session r4g: whole GPU suite, Stacking permutation soaks on the shipped build (and the fence-less build as the control), then the round-4 profiles
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4g; mkdir -p $O; rm -f $O/summary.log
timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc $? $(tail -1 $O/pytest_gpu.log)" >> $O/summary.log
for seed in 5 7; do timeout 900 python tools/gpu_stack_perm.py 8192 300 $seed > $O/soak_$seed.log 2>&1; echo "soak seed $seed rc $? $(tail -1 $O/soak_$seed.log | cut -c1-200)" >> $O/summary.log; done
D3IL_LIB_PATH=$PWD/d3il_amd/libd3il_rollout_raw.so timeout 300 python tools/gpu_stack_perm.py 2048 30 5 > $O/perm_raw.log 2>&1; echo "raw (must fail) rc $? $(tail -1 $O/perm_raw.log | cut -c1-200)" >> $O/summary.log
timeout 900 python tools/gpu_perm_push_sort.py > $O/perm_push_sort.log 2>&1; echo "perm push/sort rc $? $(tail -2 $O/perm_push_sort.log | tr '\n' ' ' | cut -c1-300)" >> $O/summary.log
cat $O/summary.log
bash tools/profile_r04.sh > $O/profile.log 2>&1
tail -25 $O/profile.log
