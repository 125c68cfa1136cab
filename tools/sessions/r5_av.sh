#!/bin/bash
# round 5 session av: Avoiding's deflate path with an extrapolated start vector and two inverse-iteration steps per shifted factorisation (ik_solve6<FAST, 1>)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5av; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_auto_reset.py tests/test_subbatch_sims.py -x -q -m gpu > $O/pytest.log 2>&1; tail -2 $O/pytest.log
timeout 600 python tools/gpu_waves.py 2>&1 | grep -v amdgpu.ids > $O/waves.log; grep "t 250\|whole run" $O/waves.log
for S in 4 1 8; do python bench.py --no-cpu-baseline --sub-batches $S 2>/dev/null | tail -1 > $O/avoiding_sb$S.json; done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]; print("%-20s %9d env-steps/s  ms %.3f  kernel %.3f" % (f.split("/")[-1][:-5], d["value"], d["ms_per_step"], r["kernel_ms"]))
PY
bash tools/probe/ik_lanes.sh 2>&1 | grep -v warning | tee $O/ik_lanes.log
