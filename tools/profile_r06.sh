#!/bin/bash
# Round-6 profiles (GPU box, repo root: bash tools/profile_r06.sh [case ...]; results in gpurun_out/r06p, the summaries copied into profiles/r06 of THIS checkout
# so that the bench lines produced afterwards carry the matching `traffic` / `valu` blocks).
# A case = task:policy:extra-bench-args.  Per case and per sub-batch count (the default of the task and 1): three rocprofv3 --pmc passes of the SAME bench
# command (FETCH_SIZE; WRITE_SIZE; the VALU group - never together, never with a trace domain, restricted to the step kernel) summarised into
# profiles/r06/pmc/<task>_<policy>_sb<S>.json; then --kernel-trace --stats of the default command, then the bench lines (default S with the cpu_baseline leg, S = 1).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06p; mkdir -p $O profiles/r06/pmc
CASES=${@:-"avoiding:random: pushing:mlp: pushing:scripted_push:--steps=100 sorting:mlp: sorting:scripted_push:--steps=60 sorting:ddpm: inserting:scripted_push:--steps=60,--warmup=5,--preroll=300 stacking:scripted_stack:--steps=100,--warmup=5 aligning:scripted_align:--steps=200,--warmup=5"}
declare -A KERN=( [avoiding]=k_avoiding_step_split [pushing]=k_sorting_step [sorting]=k_sorting_step [inserting]=k_sorting_step [stacking]=k_stacking_step [aligning]=k_aligning_step )
declare -A DEFS=( [avoiding]=4 [pushing]=4 [sorting]=4 [inserting]=4 [stacking]=1 [aligning]=1 )      # (bench.py itself defaults to 1 for the ddpm / beso policies)
for C in $CASES; do
  T=${C%%:*}; R=${C#*:}; P=${R%%:*}; X=${R#*:}; X=${X//,/ }
  K=${KERN[$T]}; B=""; if [ $T = stacking ] || [ $T = aligning ]; then B="--bimodal"; fi
  PA="--policy $P"; if [ $T = avoiding ]; then PA=""; fi
  for S in ${DEFS[$T]} 1; do
    if [ -f profiles/r06/pmc/${T}_${P}_sb$S.json ] && [ -z "$FORCE" ]; then continue; fi
    CMD="python bench.py --task $T $PA $X --sub-batches $S --no-cpu-baseline"
    PX="$X"; if [ -z "$X" ]; then PX="--steps 40"; fi            # counter passes: fewer timed steps are enough (same pre-roll, same phase mix)
    PCMD="python bench.py --task $T $PA $PX --sub-batches $S --no-cpu-baseline"
    timeout 900 rocprofv3 --kernel-include-regex "$K" --pmc FETCH_SIZE --output-format csv -d $O/pa -- $PCMD > $O/pmc_a.log 2>&1
    timeout 900 rocprofv3 --kernel-include-regex "$K" --pmc WRITE_SIZE --output-format csv -d $O/pw -- $PCMD > $O/pmc_w.log 2>&1
    timeout 900 rocprofv3 --kernel-include-regex "$K" --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $O/pb -- $PCMD > $O/pmc_b.log 2>&1
    python tools/pmc_summarize.py $B $K $O/${T}_${P}_sb$S.json $O/pa $O/pw $O/pb && cp $O/${T}_${P}_sb$S.json profiles/r06/pmc/
    rm -rf $O/pa $O/pw $O/pb
    if [ $S = 1 ] && [ ${DEFS[$T]} != 1 ]; then $CMD 2>/dev/null | tail -1 > $O/bench_line_${T}_${P}_sb1.json; fi
  done
  S=${DEFS[$T]}
  if [ ! -f $O/kernel_stats_${T}_${P}.csv ] || [ -n "$FORCE" ]; then
    timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --task $T $PA $X --no-cpu-baseline > $O/prof.log 2>&1
    f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_${T}_${P}.csv; rm -rf $O/prof
  fi
  CB="--no-cpu-baseline"; if [ $P = random ] || [ $P = mlp ] || [ $P = scripted_stack ] || [ $P = scripted_align ]; then CB=""; fi      # one cpu_baseline leg per task
  if [ $T = inserting ]; then CB=""; fi
  python bench.py --task $T $PA $X $CB 2>/dev/null | tail -1 > $O/bench_line_${T}_${P}.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06p/bench_line_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']; a=r['algorithmic_bytes_per_launch']
        print('%-44s %.3fM  ms %.3f  kernel %.3f  traffic %s (x%.1f)  isolated %s  valu %s  cpu %s  %s' % (f.split('/')[-1][11:-5], d['value']/1e6, d['ms_per_step'], r['kernel_ms'],
              r['traffic'] and round(r['traffic']/1e6,2), (r['traffic'] or 0)/a, r.get('traffic_isolated') and round(r['traffic_isolated']/1e6,2), (r['valu'] or {}).get('valu_active_frac_of_wave_cycles'),
              (d.get('cpu_baseline') or {}).get('value'), d['config'].get('flagged_envs')))
    except Exception as e: print(f, 'ERR', e)
PY
