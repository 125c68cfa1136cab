cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06w; mkdir -p $O
python bench.py --task stacking --policy beso --steps 40 --warmup 5 --no-cpu-baseline 2>$O/err_stacking_beso.log | tail -1 > $O/bench_line_stacking_beso.json
tail -3 $O/err_stacking_beso.log; python -c "
import json; d=json.loads(open('$O/bench_line_stacking_beso.json').read()); print('beso', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d.get('policy_roofline'))"
bash tools/sessions/r6v.sh
