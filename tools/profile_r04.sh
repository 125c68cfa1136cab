# Round-4 profiles (run on the GPU box from the repo root: bash tools/profile_r04.sh [tasks...]; results in gpurun_out/r04p, copied to profiles/r04).
# Per task, at the bench.py defaults (4096 envs, steady-state episode-phase mix): FIRST the four PMC counter groups (separate runs of the same
# command, never combined with a trace domain, restricted to the step kernel) and their summary - placed into profiles/r04 of THIS checkout so that
# the bench lines produced afterwards carry the matching `traffic` / `valu` blocks -, then rocprofv3 --kernel-trace --stats,
# then the bench line WITH the cpu_baseline leg, then the contact-regime / policy lines of the task.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04p; mkdir -p $O profiles/r04
TASKS=${@:-"avoiding pushing sorting stacking aligning"}
for T in $TASKS; do
  X=""; K=k_${T}_step
  if [ $T = stacking ]; then X="--steps 100 --warmup 5"; fi
  if [ $T = aligning ]; then X="--steps 200 --warmup 5"; fi
  if [ $T = inserting ]; then X="--steps 100 --warmup 5 --preroll 300"; K=k_sorting_step; fi      # the Sorting kernels with the Inserting model; 2000-step episodes: a 300-step pre-roll
  N=$K; if [ $T = avoiding ]; then N=k_avoiding_step_split; fi; if [ $T = pushing ]; then N=k_pushing_step_split; fi
  F=pmc_summary_$T.json; if [ $T = avoiding ]; then F=pmc_summary_bench300.json; fi
  B=""; if [ $T = stacking ] || [ $T = aligning ]; then B="--bimodal"; fi
  timeout 900 rocprofv3 --kernel-include-regex "$K" --pmc FETCH_SIZE --output-format csv -d $O/pmc_a_$T -- python bench.py --task $T $X --no-cpu-baseline > $O/pmc_a_$T.log 2>&1
  timeout 900 rocprofv3 --kernel-include-regex "$K" --pmc WRITE_SIZE --output-format csv -d $O/pmc_w_$T -- python bench.py --task $T $X --no-cpu-baseline > $O/pmc_w_$T.log 2>&1
  timeout 900 rocprofv3 --kernel-include-regex "$K" --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $O/pmc_b_$T -- python bench.py --task $T $X --no-cpu-baseline > $O/pmc_b_$T.log 2>&1
  timeout 900 rocprofv3 --kernel-include-regex "$K" --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/pmc_c_$T -- python bench.py --task $T $X --no-cpu-baseline > $O/pmc_c_$T.log 2>&1
  python tools/pmc_summarize.py $B $N $O/$F $O/pmc_a_$T $O/pmc_w_$T $O/pmc_b_$T $O/pmc_c_$T && cp $O/$F profiles/r04/$F
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$T -- python bench.py --task $T $X --no-cpu-baseline > $O/prof_$T.log 2>&1
  f=$(find $O/prof_$T -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_$T.csv
  rm -rf $O/prof_$T $O/pmc_a_$T $O/pmc_w_$T $O/pmc_b_$T $O/pmc_c_$T
  if [ $T = avoiding ]; then python bench.py 2>/dev/null | tail -1 > $O/bench_line_$T.json; python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_line_${T}_20steps.json
  else python bench.py --task $T $X 2>/dev/null | tail -1 > $O/bench_line_$T.json; fi
  if [ $T = pushing ] || [ $T = sorting ]; then python bench.py --task $T --policy scripted_push --steps 100 2>/dev/null | tail -1 > $O/bench_line_${T}_scripted_push.json; fi
  if [ $T = sorting ]; then python bench.py --task sorting --policy ddpm 2>/dev/null | tail -1 > $O/bench_line_sorting_ddpm.json; fi
  if [ $T = stacking ]; then python bench.py --task stacking --policy beso --steps 40 --warmup 5 --preroll 200 2>/dev/null | tail -1 > $O/bench_line_stacking_beso.json; fi
  if [ $T = aligning ]; then python bench.py --task aligning --policy mlp --steps 200 --warmup 5 2>/dev/null | tail -1 > $O/bench_line_aligning_mlp.json; fi
  if [ $T = inserting ]; then python bench.py --task inserting --policy scripted_push --steps 100 --warmup 5 --preroll 300 2>/dev/null | tail -1 > $O/bench_line_inserting_scripted_push.json; fi
done
ls -la $O
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04p/bench_line_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']; print(f.split('/')[-1], '%.3fM'%(d['value']/1e6), 'ms %.3f'%d['ms_per_step'], 'kernel %.3f'%r['kernel_ms'], 'traffic', r['traffic'], 'valu', (r['valu'] or {}).get('valu_active_frac_of_wave_cycles'), 'cpu', (d.get('cpu_baseline') or {}).get('value'), d['config'].get('flagged_envs'))
    except Exception as e: print(f, 'ERR', e)
PY
