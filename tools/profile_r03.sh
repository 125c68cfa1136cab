# Round-3 profiles: bench lines, rocprofv3 kernel-trace stats and PMC passes (separate runs) of every task's step kernel at the bench.py
# defaults (4096 envs, steady-state episode-phase mix), contact-regime lines, Avoiding batch-size sweep.  Run on the GPU box from the repo
# root: bash tools/profile_r03.sh [tasks...]; results in gpurun_out/r03p (copied to profiles/r03 afterwards).
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03p; mkdir -p $O
TASKS=${@:-"avoiding pushing sorting stacking"}
for T in $TASKS; do
  X=""; K=k_${T}_step
  if [ $T = stacking ]; then X="--steps 100 --warmup 5"; fi
  if [ $T = avoiding ]; then python bench.py 2>/dev/null | tail -1 > $O/bench_line_$T.json; python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_line_${T}_20steps.json
  else python bench.py --task $T $X --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_line_$T.json; fi
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$T -- python bench.py --task $T $X --no-cpu-baseline > $O/prof_$T.log 2>&1
  timeout 900 rocprofv3 --kernel-include-regex "$K" --pmc FETCH_SIZE --output-format csv -d $O/pmc_a_$T -- python bench.py --task $T $X --no-cpu-baseline > $O/pmc_a_$T.log 2>&1
  timeout 900 rocprofv3 --kernel-include-regex "$K" --pmc WRITE_SIZE --output-format csv -d $O/pmc_w_$T -- python bench.py --task $T $X --no-cpu-baseline > $O/pmc_w_$T.log 2>&1
  timeout 900 rocprofv3 --kernel-include-regex "$K" --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $O/pmc_b_$T -- python bench.py --task $T $X --no-cpu-baseline > $O/pmc_b_$T.log 2>&1
  timeout 900 rocprofv3 --kernel-include-regex "$K" --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/pmc_c_$T -- python bench.py --task $T $X --no-cpu-baseline > $O/pmc_c_$T.log 2>&1
  N=$K; if [ $T = avoiding ]; then N=k_avoiding_step_split; fi; if [ $T = pushing ]; then N=k_pushing_step_split; fi
  F=pmc_summary_$T.json; if [ $T = avoiding ]; then F=pmc_summary_bench300.json; fi
  B=""; if [ $T = stacking ]; then B="--bimodal"; fi
  python tools/pmc_summarize.py $B $N $O/$F $O/pmc_a_$T $O/pmc_w_$T $O/pmc_b_$T $O/pmc_c_$T
  f=$(find $O/prof_$T -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_$T.csv
  rm -rf $O/prof_$T $O/pmc_a_$T $O/pmc_w_$T $O/pmc_b_$T $O/pmc_c_$T
done
for T in pushing sorting; do
  if echo $TASKS | grep -q $T; then python bench.py --task $T --policy scripted_push --steps 100 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_line_${T}_scripted_push.json; fi
done
if echo $TASKS | grep -q avoiding; then
  for N in 4096 8192 16384 32768 65536 131072 262144; do python bench.py --envs $N --steps 120 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1; done > $O/avoiding_batch_sweep.jsonl
fi
ls -la $O
# BASELINE configs 4 / 5 with their own policies (fixed random weights), and the Stacking phase timers (diagnostics build)
if echo $TASKS | grep -q sorting; then python bench.py --task sorting --policy ddpm --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_line_sorting_ddpm.json; fi
if echo $TASKS | grep -q stacking; then
  python bench.py --task stacking --policy beso --steps 40 --warmup 5 --preroll 200 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_line_stacking_beso.json
  D3IL_STATS_LIB=1 python tools/gpu_stack_phases.py 4096 > $O/stacking_phases.log 2>&1
  python tools/gpu_beso_profile.py > $O/beso_policy_profile.log 2>&1
  for S in 5 11; do python tools/gpu_stack_perm.py 8192 300 $S 2>&1 | grep "^lib"; done > $O/stacking_permutation_soak.log
  for N in 1024 4096 8192 16384 32768; do python bench.py --task stacking --envs $N --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1; done > $O/stacking_batch_sweep.jsonl
fi
