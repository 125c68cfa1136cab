#!/bin/bash
# GPU session r4b: ISA-level patches of the RAW Stacking build (same register allocation, only waits / nops inserted), Sorting onset analysis
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4b; mkdir -p $O
for v in asm_none asm_bperm_wait asm_bperm_pre asm_vm0 asm_lgkm0 asm_dpp_nop asm_exec_nop asm_lane_nop asm_acc_nop asm_trans_nop; do
  D3IL_LIB_PATH=$PWD/d3il_amd/libd3il_rollout_$v.so timeout 600 python tools/gpu_stack_perm.py 2048 30 5 > $O/perm_$v.log 2>&1; echo "perm $v rc $?" >> $O/summary.log; tail -1 $O/perm_$v.log >> $O/summary.log
done
timeout 1500 python tools/gpu_count_onset_sorting.py --ctx 52,51,53,0,7,21 --out $O/onset_sorting.json > $O/onset_sorting.log 2>&1; echo "onset sorting rc $?" >> $O/summary.log
cat $O/summary.log; tail -20 $O/onset_sorting.log
