# round 6: BASELINE config 5 (Stacking + BESO) as two / four sub-batches: the physics launch of one sub-batch next to the policy kernels of another
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06r; mkdir -p $O
for S in 2 4; do
python bench.py --task stacking --policy beso --steps 40 --warmup 5 --no-cpu-baseline --sub-batches $S 2>$O/beso_sb$S.err | tail -1 > $O/bench_line_stacking_beso_sb$S.json
tail -3 $O/beso_sb$S.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06r/bench_line_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['value'], d['ms_per_step'], d['roofline'].get('kernel_ms'), d['config'].get('flagged_envs'))
    except Exception as e: print(f, 'ERR', e)
PY
