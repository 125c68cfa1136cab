# round 6: validation of the wrench-form Stacking / Aligning solver - permutation soaks, whole GPU suite, bench lines incl. BASELINE config 5 with its BESO policy
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06q; mkdir -p $O
for s in 5 6 7; do python tools/gpu_stack_perm.py 8192 300 $s 2>&1 | grep -v amdgpu.ids | tail -2; done | tee $O/stacking_permutation_soak.log
python bench.py --task stacking --policy scripted_stack --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_line_stacking_scripted_stack.json
python bench.py --task aligning --policy scripted_align --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_line_aligning_scripted_align.json
python bench.py --task stacking --policy beso --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_line_stacking_beso.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06q/bench_line_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['value'], d['ms_per_step'], d['roofline'].get('kernel_ms'), d['config'].get('flagged_envs'))
PY
timeout 2400 python -m pytest tests -q -m gpu -x > $O/gpu_suite.log 2>&1; tail -3 $O/gpu_suite.log
