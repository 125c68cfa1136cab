# round 6, session i: the joint solver by all 64 lanes of the wave - parity / permutation suites, bench lines of the contact regimes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06i; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity_sorting.py tests/test_gpu_parity_pushing.py tests/test_gpu_parity_inserting.py tests/test_gpu_permutation.py tests/test_sorting_sim_gpu.py tests/test_pushing_sim_gpu.py -q -m gpu -x > $O/parity.log 2>&1; grep -E "passed|failed|error" $O/parity.log | tail -3; grep -E "^FAILED|^ERROR" $O/parity.log | head
for c in "sorting scripted_push --steps=60" "inserting scripted_push --steps=60,--warmup=5,--preroll=300" "pushing scripted_push --steps=100" "sorting mlp "; do set -- $c; T=$1; P=$2; X=${3//,/ }
 for S in 4 1; do python bench.py --task $T --policy $P $X --sub-batches $S --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_line_${T}_${P}_sb$S.json; python -c "
import json; d=json.loads(open('$O/bench_line_${T}_${P}_sb$S.json').read()); print('$T $P S=$S', round(d['value']/1e6,3), 'M  ms', round(d['ms_per_step'],3), 'kernel', round(d['roofline']['kernel_ms'],3), d['config']['flagged_envs'])"; done; done
