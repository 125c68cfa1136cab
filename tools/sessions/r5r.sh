#!/bin/bash
# round 5 session r: Pushing on the generic engine - the Pushing GPU suites, bench lines of both engines (D3IL_PUSH_ENGINE=legacy: the round-1 kernel)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5r; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_parity_pushing.py tests/test_pushing_sim_gpu.py tests/test_gpu_permutation.py tests/test_gpu_auto_reset.py tests/test_gpu_full_episode_flags.py tests/test_subbatch_sims.py tests/test_gpu_count_parity.py -x -q -m gpu > $O/pytest.log 2>&1; tail -6 $O/pytest.log
for E in generic legacy; do
  export D3IL_PUSH_ENGINE=$E
  python bench.py --task pushing --no-cpu-baseline 2>/dev/null | tail -1 > $O/pushing_mlp_sb4_$E.json
  python bench.py --task pushing --no-cpu-baseline --sub-batches 1 2>/dev/null | tail -1 > $O/pushing_mlp_sb1_$E.json
  python bench.py --task pushing --policy scripted_push --steps 100 --no-cpu-baseline 2>/dev/null | tail -1 > $O/pushing_scripted_sb4_$E.json
  python bench.py --task pushing --policy scripted_push --steps 100 --no-cpu-baseline --sub-batches 1 2>/dev/null | tail -1 > $O/pushing_scripted_sb1_$E.json
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]; print("%-36s %9d env-steps/s  ms %.3f  kernel %.3f  %s  %s" % (f.split("/")[-1][:-5], d["value"], d["ms_per_step"], r["kernel_ms"], r["kernel"], d["config"]["flagged_envs"]))
    except Exception as e: print(f, "ERR", e)
PY
