#!/bin/bash
# GPU session r4c: which register-allocation component the RAW Stacking defect follows; the outlined-MPR build: soak + speed
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4c; mkdir -p $O
for v in rawwwmbasic rawwwmfast rawsgprbasic rawvgprbasic rawnodce rawnolicmflags noinl; do
  D3IL_LIB_PATH=$PWD/d3il_amd/libd3il_rollout_$v.so timeout 600 python tools/gpu_stack_perm.py 2048 30 5 > $O/perm_$v.log 2>&1; echo "perm $v rc $?" >> $O/summary.log; tail -1 $O/perm_$v.log >> $O/summary.log
done
for v in noinl rawnoinl; do
  D3IL_LIB_PATH=$PWD/d3il_amd/libd3il_rollout_$v.so timeout 900 python tools/gpu_stack_perm.py 8192 300 7 > $O/soak_$v.log 2>&1; echo "soak $v rc $?" >> $O/summary.log; tail -1 $O/soak_$v.log >> $O/summary.log
done
timeout 900 python bench.py --task stacking --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_stacking_default.json 2> $O/bench_stacking_default.err; echo "bench default rc $?" >> $O/summary.log
D3IL_LIB_PATH=$PWD/d3il_amd/libd3il_rollout_noinl.so timeout 900 python bench.py --task stacking --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_stacking_noinl.json 2> $O/bench_stacking_noinl.err; echo "bench noinl rc $?" >> $O/summary.log
D3IL_LIB_PATH=$PWD/d3il_amd/libd3il_rollout_rawnoinl.so timeout 900 python bench.py --task stacking --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_stacking_rawnoinl.json 2> $O/bench_stacking_rawnoinl.err; echo "bench rawnoinl rc $?" >> $O/summary.log
cat $O/summary.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4c/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])
    except Exception as e: print(f, 'ERR', e)
PY
