# round 6: is the intermittent Aligning failure the reordered controller phase?  The Aligning parity file ten times per library
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06ai; mkdir -p $O
for i in 1 2 3 4 5 6 7 8 9 10; do python -m pytest tests/test_gpu_parity_aligning.py -q -m gpu 2>&1 | tail -1; done | tee $O/aligning_repeat_restored.log
export D3IL_LIB_PATH=$PWD/d3il_amd/libd3il_rollout_reorder.so
for i in 1 2 3 4 5 6 7 8 9 10; do python -m pytest tests/test_gpu_parity_aligning.py -q -m gpu 2>&1 | tail -1; done | tee $O/aligning_repeat_reordered.log
