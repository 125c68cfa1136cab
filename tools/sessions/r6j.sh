cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06j; mkdir -p $O
python tools/gpu_sort_phases.py --per-wave 4096 55,90 2>&1 | grep -v amdgpu.ids | cut -c1-600 | tee $O/sort_phases_per_wave.log
