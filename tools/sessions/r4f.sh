#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4f; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity_aligning.py -x -q -s > $O/pytest_align.log 2>&1; echo "pytest align rc $?"
tail -40 $O/pytest_align.log
