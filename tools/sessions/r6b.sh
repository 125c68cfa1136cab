# round 6, session b: the tree without the legacy Pushing engines - whole GPU suite (incl. the new sub-batch / nccl readiness tests), permutation soak of the
# Stacking kernel (its code generation moved: 741 -> 716 SGPR spills), smoke, default bench line
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06b; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/gpu_suite.log 2>&1; grep -E "passed|failed|error" $O/gpu_suite.log | tail -3; grep -E "^FAILED|^ERROR" $O/gpu_suite.log | head
timeout 1200 python tools/gpu_stack_perm.py 8192 300 > $O/stack_perm.log 2>&1; tail -4 $O/stack_perm.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -8 | tee $O/smoke.log
python bench.py 2>/dev/null | tail -1 > $O/bench_default.json; python -c "
import json; d=json.loads(open('$O/bench_default.json').read()); r=d['roofline']; print(d['metric'], d['value'], d['ms_per_step'], r['frac'], r['frac_per_launch'])"
