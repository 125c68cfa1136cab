# round 6, session c: D3IL_NOINLINE = noinline + not_tail_called (no callee-saved register block around the out-of-line phases) - parity of the engines, the
# Stacking soak (its out-of-line collision phase changed too), traffic + bench lines of the contact regimes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06c; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity_sorting.py tests/test_gpu_parity_pushing.py tests/test_gpu_parity_inserting.py tests/test_gpu_permutation.py tests/test_gpu_parity_stacking.py tests/test_gpu_parity_aligning.py tests/test_gpu_parity.py -q -m gpu -x > $O/parity.log 2>&1; grep -E "passed|failed|error" $O/parity.log | tail -3; grep -E "^FAILED|^ERROR" $O/parity.log | head
timeout 900 python tools/gpu_stack_perm.py 8192 300 > $O/stack_perm.log 2>&1; tail -2 $O/stack_perm.log
export FORCE=1
bash tools/profile_r06.sh "sorting:scripted_push:--steps=60" "pushing:scripted_push:--steps=100" "sorting:mlp:" 2>&1 | tail -12
cp -r gpurun_out/r06p $O/ 2>/dev/null; cp -r profiles/r06/pmc $O/pmc 2>/dev/null
