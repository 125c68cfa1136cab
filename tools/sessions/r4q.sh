#!/bin/bash
# round 4 session q: is the HBM write traffic of the Sorting step an L2-capacity effect?  WRITE_SIZE per launch at 1024 / 2048 / 4096 environments in ONE launch per step
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4q; mkdir -p $O
for N in 1024 2048 4096; do
  timeout 600 rocprofv3 --kernel-include-regex "k_sorting_step" --pmc WRITE_SIZE --output-format csv -d $O/pmc_w_$N -- python bench.py --task sorting --envs $N --sub-batches 1 --steps 60 --no-cpu-baseline > $O/pmc_w_$N.log 2>&1
done
python - <<PY
import glob,csv,os
for d in sorted(glob.glob("$O/pmc_w_*")):
    if not os.path.isdir(d): continue
    tot=0; ids=set()
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            tot+=float(row["Counter_Value"]); ids.add(row["Dispatch_Id"])
    n=int(d.split("_")[-1]); per=tot/max(1,len(ids))
    print("Sorting, %d envs in one launch: WRITE_SIZE %.0f KiB per launch = %.1f KiB per environment (%d launches)" % (n, per, per/n, len(ids)))
PY
rm -rf $O/pmc_w_1024 $O/pmc_w_2048 $O/pmc_w_4096
