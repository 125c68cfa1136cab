#!/bin/bash
# round 5 session fd2: k_ddpm_mlp_f32 with the next tile's weights prefetched and four accumulators - parity, time per call, bench line
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5fd2; mkdir -p $O
timeout 900 python -m pytest tests/test_subbatch_sims.py tests/test_policies.py tests/test_sims_with_native_policies_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for S in 4 1; do python bench.py --task sorting --policy ddpm --no-cpu-baseline --sub-batches $S 2>/dev/null | tail -1 > $O/sorting_ddpm_sb$S.json; done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --task sorting --policy ddpm --no-cpu-baseline --sub-batches 4 > $O/prof.log 2>&1; f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_sorting_ddpm.csv; rm -rf $O/prof; grep ddpm $O/kernel_stats_sorting_ddpm.csv | cut -c1-30,180-330
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]; print("%-30s %9d env-steps/s  ms %.3f  kernel %.3f" % (f.split("/")[-1][:-5], d["value"], d["ms_per_step"], r["kernel_ms"]))
PY
