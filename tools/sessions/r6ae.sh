cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06ae; mkdir -p $O
D3IL_LIB_PATH=$PWD/d3il_amd/libd3il_rollout_noik.so python bench.py --task aligning --policy scripted_align --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_line_aligning_noik.json
python bench.py --task aligning --policy scripted_align --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_line_aligning.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06ae/bench_line_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['value'], d['ms_per_step'], d['roofline'].get('kernel_ms'))
PY
