#!/bin/bash
# round 4 session l: LDL factors handed to the serving wave (Avoiding), bench.py --task inserting
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4l; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
for i in 1 2; do python bench.py --no-cpu-baseline > $O/bench_avoiding_$i.json 2>$O/err.log; done
python bench.py --task inserting --steps 60 --warmup 5 --preroll 150 --no-cpu-baseline > $O/bench_inserting_mlp.json 2>$O/err_ins.log; tail -2 $O/err_ins.log
python bench.py --task inserting --policy scripted_push --steps 60 --warmup 5 --preroll 300 --no-cpu-baseline > $O/bench_inserting_scripted.json 2>>$O/err_ins.log; tail -2 $O/err_ins.log
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]; print(f.split("/")[-1], round(d["value"]), "env-steps/s, ms", round(d["ms_per_step"],3), "kernel", round(r["kernel_ms"],3), r.get("kernel_ms_min"), r.get("kernel_ms_max"), d["config"]["flagged_envs"], d["config"]["episodes_finished_all_ranks"], d["config"]["episodes_success_all_ranks"])
    except Exception as e: print(f, "ERR", e)
PY
