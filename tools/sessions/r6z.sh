# round 6: the wrench-form solver at BASELINE config 5's total size (32768 environments) and a second long soak; Aligning permutation at 8192
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06z; mkdir -p $O
python tools/gpu_stack_perm.py 32768 100 11 2>&1 | grep -v amdgpu.ids | tail -1 | tee $O/stacking_permutation_32768.log
python tools/gpu_stack_perm.py 8192 400 12 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $O/stacking_permutation_32768.log
python tools/gpu_stack_eval.py 2>&1 | grep -v amdgpu.ids | tail -6 | tee $O/stack_eval.log
