#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/bisect; mkdir -p $O; rm -f $O/summary.log
D3IL_LIB_PATH=$PWD/d3il_amd/libd3il_rollout_bis6862.so timeout 300 python tools/gpu_stack_perm.py 2048 30 5 > $O/perm_6862.log 2>&1; echo "6862 rc $? $(tail -1 $O/perm_6862.log | cut -c1-140)" >> $O/summary.log
D3IL_LIB_PATH=$PWD/d3il_amd/libd3il_rollout_bis13725.so timeout 300 python tools/gpu_stack_perm.py 2048 30 5 > $O/perm_13725.log 2>&1; echo "13725 rc $? $(tail -1 $O/perm_13725.log | cut -c1-140)" >> $O/summary.log
D3IL_LIB_PATH=$PWD/d3il_amd/libd3il_rollout_bis20588.so timeout 300 python tools/gpu_stack_perm.py 2048 30 5 > $O/perm_20588.log 2>&1; echo "20588 rc $? $(tail -1 $O/perm_20588.log | cut -c1-140)" >> $O/summary.log
D3IL_LIB_PATH=$PWD/d3il_amd/libd3il_rollout_bis27451.so timeout 300 python tools/gpu_stack_perm.py 2048 30 5 > $O/perm_27451.log 2>&1; echo "27451 rc $? $(tail -1 $O/perm_27451.log | cut -c1-140)" >> $O/summary.log
D3IL_LIB_PATH=$PWD/d3il_amd/libd3il_rollout_bis34314.so timeout 300 python tools/gpu_stack_perm.py 2048 30 5 > $O/perm_34314.log 2>&1; echo "34314 rc $? $(tail -1 $O/perm_34314.log | cut -c1-140)" >> $O/summary.log
D3IL_LIB_PATH=$PWD/d3il_amd/libd3il_rollout_bis41177.so timeout 300 python tools/gpu_stack_perm.py 2048 30 5 > $O/perm_41177.log 2>&1; echo "41177 rc $? $(tail -1 $O/perm_41177.log | cut -c1-140)" >> $O/summary.log
D3IL_LIB_PATH=$PWD/d3il_amd/libd3il_rollout_bis48040.so timeout 300 python tools/gpu_stack_perm.py 2048 30 5 > $O/perm_48040.log 2>&1; echo "48040 rc $? $(tail -1 $O/perm_48040.log | cut -c1-140)" >> $O/summary.log
cat $O/summary.log
