#!/bin/bash
# round 5 session pg2: policies.CapturedPolicy - tests, then the push-task bench lines with and without it
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5pg; mkdir -p $O
timeout 900 python -m pytest tests/test_subbatch_sims.py tests/test_sims_with_native_policies_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; tail -5 $O/pytest.log
for G in 0 1; do
  python bench.py --task sorting --policy ddpm --no-cpu-baseline --sub-batches 4 --policy-graph $G 2>/dev/null | tail -1 > $O/sorting_ddpm_g${G}_sb4.json
  python bench.py --task sorting --no-cpu-baseline --policy-graph $G 2>/dev/null | tail -1 > $O/sorting_mlp_g${G}_sb4.json
  python bench.py --task pushing --no-cpu-baseline --policy-graph $G 2>/dev/null | tail -1 > $O/pushing_mlp_g${G}_sb4.json
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]; print("%-36s %9d env-steps/s  ms %.3f  kernel %.3f" % (f.split("/")[-1][:-5], d["value"], d["ms_per_step"], r["kernel_ms"]))
    except Exception as e: print(f, "ERR", str(e)[:80])
PY
