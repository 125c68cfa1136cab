# round 6, fourth session: baseline of the current tree before the Stacking solver work - phase timers, bench lines, GPU suite
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06m; mkdir -p $O
D3IL_STATS_LIB=1 python tools/gpu_stack_phases.py 4096 2>&1 | grep -v amdgpu.ids | tee $O/stack_phases_before.log
python bench.py --task stacking --policy scripted_stack --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_line_stacking_scripted_stack_before.json
python bench.py --task aligning --policy scripted_align --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_line_aligning_scripted_align_before.json
python bench.py --task stacking --policy beso --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_line_stacking_beso_before.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06m/bench_line_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['value'], d['ms_per_step'], d['roofline'].get('kernel_ms'))
PY
timeout 2400 python -m pytest tests -q -m gpu -x > $O/gpu_suite.log 2>&1; tail -3 $O/gpu_suite.log
