# round 6: the cooperative-engine tests on further boxes of the pool (each gpurun call is a fresh box): product and poison library, one soak
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06am; mkdir -p $O
T=$(date +%s)
python -m pytest tests/test_gpu_parity_aligning.py tests/test_gpu_parity_stacking.py tests/test_gpu_permutation.py tests/test_gpu_poison_build.py tests/test_subbatch_sims.py -q -m gpu 2>&1 | grep -E "passed|failed" | tee $O/box_$T.log
python tools/gpu_stack_perm.py 8192 100 $((T % 97)) 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $O/box_$T.log
