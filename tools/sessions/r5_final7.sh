# round 5, last closing session: the stand-in MLP lines again (their policy now runs as k_resmlp_f32), whole GPU suite, smoke, default bench line
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05p
export FORCE=1
bash tools/profile_r05.sh "sorting:mlp:" "pushing:mlp:" 2>&1 | tail -6
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r05p/gpu_suite.log 2>&1; grep -E "passed|failed" gpurun_out/r05p/gpu_suite.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -8 | tee gpurun_out/r05p/smoke.log
python bench.py 2>/dev/null | tail -1 > gpurun_out/r05p/bench_default.json; python -c "
import json; d=json.loads(open('gpurun_out/r05p/bench_default.json').read()); print(d['metric'], d['value'], d['ms_per_step'], d['roofline']['frac'])"
