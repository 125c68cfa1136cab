# round 5, final session: per-wave lone-cube choice with expression-level contraction.  Profiles of every generic-engine case + Avoiding, the whole GPU suite,
# smoke, the permutation criterion in its long form (Pushing and Sorting), phase timers.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05p
export FORCE=1
bash tools/profile_r05.sh "avoiding:random:" "pushing:mlp:" "pushing:scripted_push:--steps=100" "sorting:mlp:" "sorting:scripted_push:--steps=60" "sorting:ddpm:" "inserting:scripted_push:--steps=60,--warmup=5,--preroll=300" 2>&1 | tail -20
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r05p/gpu_suite.log 2>&1; tail -3 gpurun_out/r05p/gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -8 | tee gpurun_out/r05p/smoke.log
for T in pushing sorting; do timeout 900 python tools/gpu_perm_push_sort.py $T 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r05p/permutation_long.log; done
python tools/gpu_gen_rest_time.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05p/rest.log
python tools/gpu_sort_phases.py 4096 55,90 2>&1 | grep "per workgroup" | tee gpurun_out/r05p/phases_4096_final.log
