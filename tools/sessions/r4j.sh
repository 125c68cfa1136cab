#!/bin/bash
# round 4 session j: Inserting on the device (generic engine + rod <-> wall contacts): its parity tests, then the Sorting tests (the engine changed under them)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4j; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity_inserting.py -x -q -m gpu -s > $O/inserting.log 2>&1; tail -25 $O/inserting.log
timeout 1500 python -m pytest tests/test_gpu_parity_sorting.py tests/test_gpu_parity.py -x -q -m gpu > $O/sorting.log 2>&1; tail -5 $O/sorting.log
