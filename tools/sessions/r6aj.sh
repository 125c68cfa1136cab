cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06aj; mkdir -p $O
python -m pytest tests/test_gpu_parity_aligning.py -q -m gpu -x 2>&1 | grep -v "^ctx60\|^  \|^align_blob" | tail -40 > $O/fail.log; grep -n "Error\|assert\|FAILED\|passed" $O/fail.log | head -20
rocm-smi --showproductname 2>/dev/null | head -8; cat /sys/class/kfd/kfd/topology/nodes/*/name 2>/dev/null | head
