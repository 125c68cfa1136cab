# round 6: the NaN-poison build (every LDS word starts as a NaN) on the Aligning and Stacking parity files; then the product build
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06ak; mkdir -p $O
export D3IL_LIB_PATH=$PWD/d3il_amd/libd3il_rollout_poison.so
python -m pytest tests/test_gpu_parity_aligning.py -q -m gpu 2>&1 | grep -E "passed|failed" | tee $O/poison_aligning.log
python -m pytest tests/test_gpu_parity_stacking.py tests/test_gpu_permutation.py -q -m gpu 2>&1 | grep -E "passed|failed" | tee $O/poison_stacking.log
unset D3IL_LIB_PATH
python -m pytest tests/test_gpu_parity_aligning.py tests/test_gpu_parity_stacking.py tests/test_gpu_permutation.py -q -m gpu 2>&1 | grep -E "passed|failed" | tee $O/product.log
