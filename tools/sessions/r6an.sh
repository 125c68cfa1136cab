cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06an; mkdir -p $O
python -m pytest tests/test_gpu_count_parity.py -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tee $O/count_parity.log
