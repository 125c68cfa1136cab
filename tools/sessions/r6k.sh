cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06k; mkdir -p $O
timeout 900 python -m pytest tests/test_policies_f16x3.py tests/test_policies.py tests/test_sims_with_native_policies_gpu.py -q -m gpu -s > $O/policy.log 2>&1; grep -E "passed|failed|error|mlp rows" $O/policy.log | tail -8
python tools/probe/f16x3_bench.py 45056 2>&1 | grep "mlp\|linear" | tee $O/f16x3_bench.log
python tools/gpu_beso_profile.py 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tee $O/beso_profile_f16x3.log | grep -E "predict_batch|f16x3|attention|Self CUDA time"
python bench.py --task stacking --policy beso --steps 40 --warmup 5 --no-cpu-baseline 2>$O/beso.err | tail -1 > $O/bench_line_stacking_beso.json
python -c "
import json; d=json.loads(open('$O/bench_line_stacking_beso.json').read()); print('beso', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
