#!/bin/bash
# round 5 session b: where the tree solver's time goes (stats build), contact regime
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5b; mkdir -p $O
python tools/gpu_sort_phases.py 1024 2>&1 | grep -v amdgpu.ids | tee $O/phases_1024.log
python tools/gpu_sort_phases.py 4096 2>&1 | grep -v amdgpu.ids | tee $O/phases_4096.log
