#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4h; mkdir -p $O
bash tools/profile_r04.sh aligning > $O/profile_align.log 2>&1; tail -6 $O/profile_align.log
# Stacking engine with two environments per workgroup (2048 workgroups, two waves per SIMD) against the shipped four
for v in lanes2; do
  D3IL_LIB_PATH=$PWD/d3il_amd/libd3il_rollout_$v.so timeout 600 python bench.py --task stacking --steps 100 --warmup 5 --no-cpu-baseline 2>$O/bench_$v.err | tail -1 > $O/bench_stacking_$v.json
  D3IL_LIB_PATH=$PWD/d3il_amd/libd3il_rollout_$v.so timeout 600 python bench.py --task aligning --steps 200 --warmup 5 --no-cpu-baseline 2>>$O/bench_$v.err | tail -1 > $O/bench_aligning_$v.json
  D3IL_LIB_PATH=$PWD/d3il_amd/libd3il_rollout_$v.so timeout 600 python tools/gpu_stack_perm.py 4096 100 5 2>&1 | tail -1 | cut -c1-200
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4h/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], '%.3fM'%(d['value']/1e6), 'ms %.3f'%d['ms_per_step'], 'kernel %.3f'%d['roofline']['kernel_ms'], d['config'].get('flagged_envs'))
    except Exception as e: print(f, 'ERR', e)
PY
