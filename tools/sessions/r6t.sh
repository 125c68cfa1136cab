# round 6: the attention half of the DiffusionGPT block as one kernel - tests, policy profile, BASELINE config 5 line
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06t; mkdir -p $O
timeout 900 python -m pytest tests/test_policies_f16x3.py tests/test_policies.py tests/test_sims_with_native_policies_gpu.py -q -m gpu -x > $O/policy_tests.log 2>&1; tail -12 $O/policy_tests.log
python tools/gpu_beso_profile.py 2>&1 | grep -v amdgpu.ids | cut -c1-200 > $O/beso_profile_attn_half.log; grep -E "predict_batch|Self CUDA time total" $O/beso_profile_attn_half.log; grep -E "k_mlp|k_linear|k_attention|k_attn" $O/beso_profile_attn_half.log | cut -c1-40,150-200
python bench.py --task stacking --policy beso --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_line_stacking_beso.json
python -c "
import json; d=json.loads(open('$O/bench_line_stacking_beso.json').read()); print('beso', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
