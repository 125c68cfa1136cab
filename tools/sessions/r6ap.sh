cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06ap; mkdir -p $O
python -m pytest tests/test_gpu_parity_stacking.py tests/test_gpu_parity_aligning.py -q -m gpu -x 2>&1 | grep -E "passed|failed" | tee $O/tests.log
D3IL_STATS_LIB=1 python tools/gpu_stack_phases.py 4096 2>&1 | grep -v amdgpu.ids | tail -4 | head -2 | cut -c1-330
python bench.py --task stacking --policy scripted_stack --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('stacking', d['value'], d['ms_per_step'])"
