cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05p
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r05p/gpu_suite.log 2>&1; tail -4 gpurun_out/r05p/gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -8 | tee gpurun_out/r05p/smoke.log
bash tools/profile_r05.sh 2>&1 | tail -30
