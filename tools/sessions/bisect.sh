#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/bisect; mkdir -p $O; rm -f $O/summary.log
for f in d3il_amd/libd3il_rollout_bis*.so; do n=$(basename $f .so | sed 's/libd3il_rollout_bis//'); D3IL_LIB_PATH=$PWD/$f timeout 300 python tools/gpu_stack_perm.py 2048 30 5 > $O/perm_$n.log 2>&1; echo "$n rc $? $(tail -1 $O/perm_$n.log | cut -c1-170)" >> $O/summary.log; done
cat $O/summary.log
