set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r01b; mkdir -p $O
for T in avoiding sorting pushing; do
  python bench.py --task $T --steps 300 --warmup 20 2>/dev/null | tail -1 > $O/bench_line_$T.json
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$T -- python bench.py --task $T --no-cpu-baseline > $O/prof_$T.log 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_a_$T -- python bench.py --task $T --no-cpu-baseline > $O/pmc_a_$T.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w_$T -- python bench.py --task $T --no-cpu-baseline > $O/pmc_w_$T.log 2>&1
  timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $O/pmc_b_$T -- python bench.py --task $T --no-cpu-baseline > $O/pmc_b_$T.log 2>&1
  timeout 600 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/pmc_c_$T -- python bench.py --task $T --no-cpu-baseline > $O/pmc_c_$T.log 2>&1
done
python tools/pmc_summarize.py k_avoiding_step_split $O/pmc_summary_avoiding.json $O/pmc_a_avoiding $O/pmc_w_avoiding $O/pmc_b_avoiding $O/pmc_c_avoiding
python tools/pmc_summarize.py k_sorting_step $O/pmc_summary_sorting.json $O/pmc_a_sorting $O/pmc_w_sorting $O/pmc_b_sorting $O/pmc_c_sorting
python tools/pmc_summarize.py k_pushing_step_split $O/pmc_summary_pushing.json $O/pmc_a_pushing $O/pmc_w_pushing $O/pmc_b_pushing $O/pmc_c_pushing
for T in avoiding sorting pushing; do f=$(find $O/prof_$T -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_$T.csv; done
timeout 300 python tools/gpu_sort_time.py 4096 70 > $O/sort_time.log 2>&1
# keep the merge small: drop the raw traces
rm -rf $O/prof_* $O/pmc_a_* $O/pmc_w_* $O/pmc_b_* $O/pmc_c_*
ls -la $O
