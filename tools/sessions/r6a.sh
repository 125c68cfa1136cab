# round 6, first session: baseline of the round-5 tree - GPU suite, BASELINE config 5 with its own BESO policy (not re-measured in round 5), its kernel trace
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06a; mkdir -p $O
python bench.py --task stacking --policy beso --steps 40 --warmup 5 2>$O/beso.err | tail -1 > $O/bench_line_stacking_beso.json
python -c "
import json; d=json.loads(open('$O/bench_line_stacking_beso.json').read()); print('beso', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d.get('policy_roofline'))"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --task stacking --policy beso --steps 20 --warmup 3 --no-cpu-baseline > $O/prof.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_stacking_beso.csv; rm -rf $O/prof
head -25 $O/kernel_stats_stacking_beso.csv
timeout 2400 python -m pytest tests -q -m gpu -x > $O/gpu_suite.log 2>&1; tail -3 $O/gpu_suite.log
