cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06ac; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity_aligning.py tests/test_gpu_permutation.py -q -m gpu -x > $O/tests.log 2>&1; tail -2 $O/tests.log
python bench.py --task aligning --policy scripted_align --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_line_aligning_scripted_align.json
python bench.py --task stacking --policy scripted_stack --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_line_stacking_scripted_stack.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06ac/bench_line_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['value'], d['ms_per_step'], d['roofline'].get('kernel_ms'), d['config'].get('flagged_envs'))
PY
