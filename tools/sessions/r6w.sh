# round 6, final tree: the bench lines of every regime once more, now with the committed counter summaries AND the counted FP64 mix in place (valu block), the default invocation, config 5
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06w; mkdir -p $O
python bench.py 2>/dev/null | tail -1 > $O/bench_line_default_invocation.json
run() { n=$1; shift; python bench.py "$@" 2>$O/err_$n.log | tail -1 > $O/bench_line_$n.json; }
run avoiding_random --task avoiding
run avoiding_random_sb1 --task avoiding --sub-batches 1 --no-cpu-baseline
run pushing_mlp --task pushing --policy mlp
run pushing_mlp_sb1 --task pushing --policy mlp --sub-batches 1 --no-cpu-baseline
run pushing_scripted_push --task pushing --policy scripted_push --steps 100 --no-cpu-baseline
run pushing_scripted_push_sb1 --task pushing --policy scripted_push --steps 100 --sub-batches 1 --no-cpu-baseline
run sorting_mlp --task sorting --policy mlp
run sorting_mlp_sb1 --task sorting --policy mlp --sub-batches 1 --no-cpu-baseline
run sorting_scripted_push --task sorting --policy scripted_push --steps 60 --no-cpu-baseline
run sorting_scripted_push_sb1 --task sorting --policy scripted_push --steps 60 --sub-batches 1 --no-cpu-baseline
run sorting_ddpm --task sorting --policy ddpm --no-cpu-baseline
run sorting_ddpm_sb1 --task sorting --policy ddpm --sub-batches 1 --no-cpu-baseline
run inserting_scripted_push --task inserting --policy scripted_push --steps 60 --warmup 5 --preroll 300
run inserting_scripted_push_sb1 --task inserting --policy scripted_push --steps 60 --warmup 5 --preroll 300 --sub-batches 1 --no-cpu-baseline
run stacking_scripted_stack --task stacking --policy scripted_stack --steps 100 --warmup 5
run aligning_scripted_align --task aligning --policy scripted_align --steps 200 --warmup 5
run stacking_beso --task stacking --policy beso --steps 40 --warmup 5 --no-cpu-baseline
tail -5 $O/err_stacking_beso.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06w/bench_line_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print('%-40s %.3fM ms %.3f kernel %.3f frac %.2e traffic %s valu %s cpu %s' % (f.split('/')[-1][11:-5], d['value']/1e6, d['ms_per_step'], r['kernel_ms'], r['frac'], r.get('traffic') and round(r['traffic']/1e6,2), (r.get('valu') or {}).get('valu_active_frac_of_wave_cycles'), (d.get('cpu_baseline') or {}).get('value')))
    except Exception as e: print(f, 'ERR', e)
PY
