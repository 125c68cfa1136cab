# round 6: Aligning with the controller pass against the previous commit's library (in-kernel controller), same box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06ah; mkdir -p $O
for L in cur prev; do
  if [ $L = prev ]; then export D3IL_LIB_PATH=$PWD/d3il_amd/libd3il_rollout_prev.so; else unset D3IL_LIB_PATH; fi
  python bench.py --task aligning --policy scripted_align --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_line_aligning_scripted_align_$L.json
  python bench.py --task aligning --policy mlp --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_line_aligning_mlp_$L.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06ah/bench_line_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['value'], d['ms_per_step'], d['roofline'].get('kernel_ms'), d['config'].get('flagged_envs'))
PY
