#!/bin/bash
# round 4 session n: padded row stride of the SoA buffers (L2 set conflicts at 32 KiB strides) - A/B on time and on WRITE_SIZE / FETCH_SIZE
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4n; mkdir -p $O
for P in 64 0; do
  export D3IL_STRIDE_PAD=$P
  for T in sorting pushing; do
    python bench.py --task $T --no-cpu-baseline > $O/bench_${T}_pad$P.json 2>/dev/null
    python bench.py --task $T --policy scripted_push --steps 100 --no-cpu-baseline > $O/bench_${T}_scripted_pad$P.json 2>/dev/null
  done
  python bench.py --no-cpu-baseline > $O/bench_avoiding_pad$P.json 2>/dev/null
  python bench.py --task stacking --steps 100 --warmup 5 --no-cpu-baseline > $O/bench_stacking_pad$P.json 2>/dev/null
  python bench.py --task aligning --steps 100 --warmup 5 --no-cpu-baseline > $O/bench_aligning_pad$P.json 2>/dev/null
  for T in sorting pushing; do
    K=k_${T}_step
    timeout 600 rocprofv3 --kernel-include-regex "$K" --pmc WRITE_SIZE --output-format csv -d $O/pmc_w_${T}_$P -- python bench.py --task $T --steps 60 --no-cpu-baseline > $O/pmc_w_${T}_$P.log 2>&1
    timeout 600 rocprofv3 --kernel-include-regex "$K" --pmc FETCH_SIZE --output-format csv -d $O/pmc_a_${T}_$P -- python bench.py --task $T --steps 60 --no-cpu-baseline > $O/pmc_a_${T}_$P.log 2>&1
  done
done
python - <<PY
import json,glob,csv,os
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]; print(f.split("/")[-1], round(d["value"]), "env-steps/s, ms", round(d["ms_per_step"],3), "kernel", round(r["kernel_ms"],3), d["config"]["flagged_envs"])
    except Exception as e: print(f, "ERR", e)
for d in sorted(glob.glob("$O/pmc_?_*_*")):
    if not os.path.isdir(d): continue
    tot={}; n={}
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k=row["Counter_Name"]; tot[k]=tot.get(k,0)+float(row["Counter_Value"]); n[k]=n.get(k,0)+1
    # per dispatch: counters are per (dispatch, xcc...) rows; count dispatches by unique Dispatch_Id
    ids=set()
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)): ids.add(row["Dispatch_Id"])
    for k in tot: print(d.split("/")[-1], k, "KiB per dispatch: %.0f (%d dispatches)" % (tot[k]/max(1,len(ids)), len(ids)))
PY
rm -rf $O/pmc_?_*_64 $O/pmc_?_*_0
