cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06o; mkdir -p $O
D3IL_STATS_LIB=1 python tools/gpu_stack_phases.py 4096 2>&1 | grep -v amdgpu.ids | tee $O/stack_phases_wrench_sub.log
