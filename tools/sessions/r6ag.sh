cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for i in 1 2 3 4; do python -m pytest tests/test_gpu_parity_aligning.py -q -m gpu 2>&1 | tail -1; done
for i in 1 2; do python -m pytest tests/test_gpu_parity_aligning.py tests/test_gpu_permutation.py tests/test_subbatch_sims.py -q -m gpu 2>&1 | tail -1; done
