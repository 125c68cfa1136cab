# round 6, session e: where the BESO policy step goes with the split-f16 kernels (both GEMM modes), phase timers of the Sorting step (stats build), the box's compiler
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06e; mkdir -p $O
/opt/rocm/bin/hipcc --version 2>&1 | head -3 | tee $O/hipcc_version.txt
python tools/gpu_beso_profile.py 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tee $O/beso_profile_f16x3.log | head -40
D3IL_POLICY_GEMM=f32 python tools/gpu_beso_profile.py 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tee $O/beso_profile_f32.log | head -16
python tools/gpu_sort_phases.py 4096 55,90 2>&1 | grep -v amdgpu.ids | cut -c1-400 | tee $O/sort_phases.log
