#!/bin/bash
# round 4 session p: bench.py with sub-batches (default 4) - contract tests, lines of every task at 1 and 4 sub-batches
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4p; mkdir -p $O
timeout 1500 python -m pytest tests/test_bench_contract.py -x -q -m gpu > $O/tests.log 2>&1; tail -4 $O/tests.log
for SB in 1 4; do
  python bench.py --no-cpu-baseline --sub-batches $SB > $O/avoiding_sb$SB.json 2>$O/err.log
  python bench.py --task pushing --no-cpu-baseline --sub-batches $SB > $O/pushing_sb$SB.json 2>>$O/err.log
  python bench.py --task pushing --policy scripted_push --steps 100 --no-cpu-baseline --sub-batches $SB > $O/pushing_scripted_sb$SB.json 2>>$O/err.log
  python bench.py --task sorting --no-cpu-baseline --sub-batches $SB > $O/sorting_sb$SB.json 2>>$O/err.log
  python bench.py --task sorting --policy ddpm --no-cpu-baseline --sub-batches $SB > $O/sorting_ddpm_sb$SB.json 2>>$O/err.log
  python bench.py --task sorting --policy scripted_push --steps 100 --no-cpu-baseline --sub-batches $SB > $O/sorting_scripted_sb$SB.json 2>>$O/err.log
  python bench.py --task stacking --steps 100 --warmup 5 --no-cpu-baseline --sub-batches $SB > $O/stacking_sb$SB.json 2>>$O/err.log
  python bench.py --task stacking --policy beso --steps 40 --warmup 5 --preroll 200 --no-cpu-baseline --sub-batches $SB > $O/stacking_beso_sb$SB.json 2>>$O/err.log
  python bench.py --task aligning --steps 100 --warmup 5 --no-cpu-baseline --sub-batches $SB > $O/aligning_sb$SB.json 2>>$O/err.log
  python bench.py --task inserting --steps 100 --warmup 5 --preroll 300 --no-cpu-baseline --sub-batches $SB > $O/inserting_sb$SB.json 2>>$O/err.log
done
tail -3 $O/err.log
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*_sb*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]; c=d["config"]; print("%-28s %9d env-steps/s  ms %.3f  kernel %.3f (launch of %d envs)  episodes %d / %d  %s" % (f.split("/")[-1][:-5], d["value"], d["ms_per_step"], r["kernel_ms"], c["envs_per_launch"], c["episodes_finished_all_ranks"], c["episodes_success_all_ranks"], c["flagged_envs"]))
    except Exception as e: print(f, "ERR", e)
PY
