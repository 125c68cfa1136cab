# Quick check of one task's step kernel: bench line, then HBM traffic per launch (FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes - together they hang)
# and VALU utilisation.  usage (GPU box, repo root): bash tools/quick_traffic.sh stacking k_stacking_step "--steps 60 --warmup 5 --preroll 400"
T=$1; K=$2; X=$3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/quick_$T; rm -rf $O; mkdir -p $O
python bench.py --task $T $X --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-220
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_a -- python bench.py --task $T $X --no-cpu-baseline > $O/pmc_a.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python bench.py --task $T $X --no-cpu-baseline > $O/pmc_w.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $O/pmc_b -- python bench.py --task $T $X --no-cpu-baseline > $O/pmc_b.log 2>&1
python tools/pmc_summarize.py $K $O/pmc.json $O/pmc_a $O/pmc_w $O/pmc_b
rm -rf $O/pmc_a $O/pmc_b $O/pmc_w
