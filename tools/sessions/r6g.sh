# round 6, session g: workgroup shapes of the split-f16 kernels (probe), per-wave phase timers + barrier waits of the Sorting step
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06g; mkdir -p $O
python tools/probe/f16x3_bench.py 45056 2>&1 | grep -v amdgpu.ids | tee $O/f16x3_bench.log
python tools/gpu_sort_phases.py --per-wave 4096 55,90 2>&1 | grep -v amdgpu.ids | cut -c1-600 | tee $O/sort_phases_per_wave.log
