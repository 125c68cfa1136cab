#!/bin/bash
# round 5 session rm: the ResidualMLPNetwork as one matrix-core kernel (k_resmlp_f32): parity, policy suites, the stand-in MLP bench lines with and without
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5rm; mkdir -p $O
timeout 900 python -m pytest tests/test_subbatch_sims.py tests/test_policies.py tests/test_sims_with_native_policies_gpu.py tests/test_pushing_sim_gpu.py tests/test_sorting_sim_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; tail -12 $O/pytest.log | grep -v "^$" | tail -8
for F in 0 1; do
  D3IL_POLICY_FUSED_RESMLP=$F python bench.py --task sorting --no-cpu-baseline 2>/dev/null | tail -1 > $O/sorting_mlp_fused$F.json
  D3IL_POLICY_FUSED_RESMLP=$F python bench.py --task pushing --no-cpu-baseline 2>/dev/null | tail -1 > $O/pushing_mlp_fused$F.json
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]; print("%-30s %9d env-steps/s  ms %.3f  kernel %.3f" % (f.split("/")[-1][:-5], d["value"], d["ms_per_step"], r["kernel_ms"]))
PY
