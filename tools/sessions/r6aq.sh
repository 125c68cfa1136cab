# round 6: last check of the rebuilt libraries of the committed tree (cooperative-engine tests on product + poison library, smoke)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06aq; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep "smoke ok" | tee $O/smoke.log
python -m pytest tests/test_gpu_parity_aligning.py tests/test_gpu_parity_stacking.py tests/test_gpu_permutation.py tests/test_gpu_poison_build.py tests/test_policies_f16x3.py -q -m gpu 2>&1 | grep -E "passed|failed" | tee $O/tests.log
python bench.py 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('default bench', d['value'], d['ms_per_step'], d['roofline']['frac'])"
