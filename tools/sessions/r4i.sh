#!/bin/bash
# round 4 session i: the three-wave Avoiding kernel (rare constraint paths in their own wave): parity tests, same-box A/B against the two-wave form
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4i; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_episode_flags.py tests/test_gpu_auto_reset.py tests/test_gpu_permutation.py -x -q -m gpu > $O/tests.log 2>&1; tail -5 $O/tests.log
for v in 256 0 256 0; do
  python bench.py --no-cpu-baseline --serve-max-wg $v > $O/bench_serve$v.json 2>$O/bench_serve$v.err; python - <<PY
import json; d=json.loads(open("$O/bench_serve$v.json").read().strip().splitlines()[-1]); r=d["roofline"]; print("serve_max_wg $v:", round(d["value"]), "env-steps/s, kernel ms", r["kernel_ms"], r.get("kernel_ms_min"), r.get("kernel_ms_max"))
PY
done
python bench.py --no-cpu-baseline --envs 65536 --steps 60 --serve-max-wg 4096 > $O/bench_65536_serve.json 2>/dev/null; python bench.py --no-cpu-baseline --envs 65536 --steps 60 > $O/bench_65536.json 2>/dev/null
python - <<PY
import json
for f in ("bench_65536_serve","bench_65536"):
    d=json.loads(open("$O/%s.json"%f).read().strip().splitlines()[-1]); print(f, round(d["value"]), d["roofline"]["kernel_ms"])
PY
