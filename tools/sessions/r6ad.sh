# round 6: Stacking / Aligning as sub-batches (the launch of a sub-batch ends with ITS slowest workgroup)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06ad; mkdir -p $O
for S in 2 4 8; do
python bench.py --task stacking --policy scripted_stack --steps 100 --warmup 5 --no-cpu-baseline --sub-batches $S 2>/dev/null | tail -1 > $O/bench_line_stacking_scripted_stack_sb$S.json
python bench.py --task aligning --policy scripted_align --steps 200 --warmup 5 --no-cpu-baseline --sub-batches $S 2>/dev/null | tail -1 > $O/bench_line_aligning_scripted_align_sb$S.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06ad/bench_line_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['value'], d['ms_per_step'], d['roofline'].get('kernel_ms'), d['config'].get('flagged_envs'))
    except Exception as e: print(f,'ERR',e)
PY
