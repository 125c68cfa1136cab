# round 6: Cholesky with the earlier blocks' row entries read from LDS - cooperative tests (product + poison), phases, bench
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06ao; mkdir -p $O
python -m pytest tests/test_gpu_parity_aligning.py tests/test_gpu_parity_stacking.py tests/test_gpu_permutation.py tests/test_gpu_poison_build.py -q -m gpu 2>&1 | grep -E "passed|failed" | tee $O/tests.log
D3IL_STATS_LIB=1 python tools/gpu_stack_phases.py 4096 2>&1 | grep -v amdgpu.ids | tail -5 | cut -c1-330 | tee $O/stack_phases.log
python bench.py --task stacking --policy scripted_stack --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_line_stacking_scripted_stack.json
python bench.py --task aligning --policy scripted_align --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_line_aligning_scripted_align.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06ao/bench_line_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['value'], d['ms_per_step'], d['roofline'].get('kernel_ms'), d['config'].get('flagged_envs'))
PY
