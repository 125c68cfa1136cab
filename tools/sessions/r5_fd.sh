#!/bin/bash
# round 5 session fd: the DDPM policy's sampling chain as one matrix-core kernel (k_ddpm_mlp_f32) - parity with the torch chain, the policy suites, bench lines
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5fd; mkdir -p $O
timeout 900 python -m pytest tests/test_subbatch_sims.py -x -q -m gpu -k "fused_ddpm or captured" > $O/pytest_fused.log 2>&1; tail -15 $O/pytest_fused.log | grep -v "^$"
timeout 900 python -m pytest tests/test_policies.py tests/test_sims_with_native_policies_gpu.py -x -q -m gpu > $O/pytest_policies.log 2>&1; tail -3 $O/pytest_policies.log
for F in 0 1; do for S in 1 4; do
  D3IL_POLICY_FUSED_DDPM=$F python bench.py --task sorting --policy ddpm --no-cpu-baseline --sub-batches $S 2>$O/err_f${F}_sb$S.log | tail -1 > $O/sorting_ddpm_fused${F}_sb$S.json
done; done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]; print("%-36s %9d env-steps/s  ms %.3f  kernel %.3f" % (f.split("/")[-1][:-5], d["value"], d["ms_per_step"], r["kernel_ms"]))
    except Exception as e: print(f, "ERR", str(e)[:80])
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --task sorting --policy ddpm --no-cpu-baseline --sub-batches 4 > $O/prof.log 2>&1; f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_sorting_ddpm_fused.csv; rm -rf $O/prof; head -4 $O/kernel_stats_sorting_ddpm_fused.csv | cut -c1-50,180-330
