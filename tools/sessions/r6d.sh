# round 6, session d: the generic engine without scratch traffic (arm state parked in LDS, contacts emitted straight into the records, the t area one contiguous
# block per environment) + the split-f16 policy kernels: parity, permutation, traffic and bench lines of the contact regimes, BASELINE config 5 with BESO
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06d; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity_sorting.py tests/test_gpu_parity_pushing.py tests/test_gpu_parity_inserting.py tests/test_gpu_permutation.py tests/test_sorting_sim_gpu.py tests/test_pushing_sim_gpu.py -q -m gpu -x > $O/parity.log 2>&1; grep -E "passed|failed|error" $O/parity.log | tail -3; grep -E "^FAILED|^ERROR" $O/parity.log | head
timeout 900 python -m pytest tests/test_policies_f16x3.py tests/test_policies.py tests/test_sims_with_native_policies_gpu.py -q -m gpu -s > $O/policy.log 2>&1; grep -E "passed|failed|error|rows" $O/policy.log | tail -16
python bench.py --task stacking --policy beso --steps 40 --warmup 5 --no-cpu-baseline 2>$O/beso.err | tail -1 > $O/bench_line_stacking_beso.json
python -c "
import json; d=json.loads(open('$O/bench_line_stacking_beso.json').read()); print('beso', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d.get('policy_roofline'))"
export FORCE=1
bash tools/profile_r06.sh "sorting:scripted_push:--steps=60" "pushing:scripted_push:--steps=100" "sorting:mlp:" "pushing:mlp:" 2>&1 | tail -10
cp -r gpurun_out/r06p $O/ 2>/dev/null; cp -r profiles/r06/pmc $O/pmc 2>/dev/null
