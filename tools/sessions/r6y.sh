# round 6: closing run of the committed tree - smoke(), whole GPU suite
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06y; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee $O/smoke.log | tail -8
timeout 2400 python -m pytest tests -q -m gpu > $O/gpu_suite.log 2>&1; tail -3 $O/gpu_suite.log
