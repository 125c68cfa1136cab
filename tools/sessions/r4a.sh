#!/bin/bash
# GPU session r4a (round 4): code-generation experiments on the RAW Stacking build, divergence-onset analysis, strict A/B, Avoiding wave times
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4a; mkdir -p $O
for v in raw rawwait rawlive rawexec rawdpp rawnoinl; do
  D3IL_LIB_PATH=$PWD/d3il_amd/libd3il_rollout_$v.so timeout 600 python tools/gpu_stack_perm.py 2048 30 5 > $O/perm_$v.log 2>&1; echo "perm $v rc $?" >> $O/summary.log; tail -1 $O/perm_$v.log >> $O/summary.log
done
timeout 600 python tools/gpu_stack_perm.py 2048 30 5 > $O/perm_default.log 2>&1; echo "perm default rc $?" >> $O/summary.log; tail -1 $O/perm_default.log >> $O/summary.log
timeout 1200 python tools/gpu_count_onset.py --ctx 10,26,34,54,6,22,2,14,1,3 --out $O/onset_pushing.json > $O/onset_pushing.log 2>&1; echo "onset rc $?" >> $O/summary.log
timeout 1200 python tools/gpu_count_onset.py --strict 1 --ctx 10,26,34,54,6,22,2,14,1,3 --out $O/onset_pushing_strict.json > $O/onset_pushing_strict.log 2>&1; echo "onset strict rc $?" >> $O/summary.log
timeout 1200 python tools/gpu_count_strict.py pushing pushing_sampled > $O/count_strict.log 2>&1; echo "strict ab rc $?" >> $O/summary.log
timeout 600 python tools/gpu_waves.py > $O/waves.log 2>&1; echo "waves rc $?" >> $O/summary.log
cat $O/summary.log; tail -14 $O/onset_pushing.log; tail -5 $O/count_strict.log; tail -30 $O/waves.log
