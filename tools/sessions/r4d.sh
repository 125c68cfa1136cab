#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4d; mkdir -p $O; rm -f $O/summary.log
for v in rawf_a rawf_b rawf_c rawf_d rawf_f rawf_g rawf_h rawf_i; do D3IL_LIB_PATH=$PWD/d3il_amd/libd3il_rollout_$v.so timeout 300 python tools/gpu_stack_perm.py 2048 30 5 > $O/perm_$v.log 2>&1; echo "$v rc $? $(tail -1 $O/perm_$v.log | cut -c1-170)" >> $O/summary.log; done
cat $O/summary.log
