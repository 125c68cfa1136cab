# round 6: permutation soaks of the generic engine on the final library (Pushing, Sorting 4096 x 250)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06ar; mkdir -p $O
for t in pushing sorting; do timeout 600 python tools/gpu_perm_push_sort.py $t 4096 250 31 2>&1 | grep -v amdgpu.ids | tail -1; done | tee $O/push_sort_permutation_soak.log
