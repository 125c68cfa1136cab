# round 6: divergence onset of every context the count-parity gate calls undecided: device against oracle, free-running and one-step
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06v; mkdir -p $O
python tools/gpu_count_onset.py --ctx 6,10,22,26,30,34,54 --out $O/onset_pushing.json 2>&1 | grep -v amdgpu.ids | tee $O/onset_pushing.log | cut -c1-200
python tools/gpu_count_onset.py --ctx 6,10,22,26,30,34,54 --strict 1 --out $O/onset_pushing_strict.json 2>&1 | grep -v amdgpu.ids | tee $O/onset_pushing_strict.log | cut -c1-200
python tools/gpu_count_onset.py --sampled --ctx 6,38,42,46,62,66,70,82,94,102,110,114 --out $O/onset_pushing_sampled.json 2>&1 | grep -v amdgpu.ids | tee $O/onset_pushing_sampled.log | cut -c1-200
python tools/gpu_count_onset_sorting.py --ctx 34,45,52 --out $O/onset_sorting.json 2>&1 | grep -v amdgpu.ids | tee $O/onset_sorting.log | cut -c1-200
