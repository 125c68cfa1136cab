#!/bin/bash
# round 5 session h: what the physics waves of the generic engine wait for - instruction cache / issue counters of k_sorting_step (resting / MLP regime)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5h; mkdir -p $O
X="--task sorting --no-cpu-baseline --sub-batches 1 --steps 30 --warmup 5 ${1:-}"
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_IFETCH SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-include-regex k_sorting_step --pmc $set --output-format csv -d $O/pmc_$i -- python bench.py $X > $O/pmc_$i.log 2>&1 || echo "set $i failed: $set"
done
python tools/pmc_summarize.py k_sorting_step $O/pmc_issue${2:-}.json $O/pmc_1 $O/pmc_2 $O/pmc_3 $O/pmc_4 $O/pmc_5
rm -rf $O/pmc_[0-9]
