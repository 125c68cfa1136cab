cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06l; mkdir -p $O
timeout 600 rocprofv3 --kernel-include-regex "k_mlp_gelu_residual_f16x3" --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/p1 -- python tools/probe/f16x3_bench.py 45056 > $O/p1.log 2>&1
timeout 600 rocprofv3 --kernel-include-regex "k_mlp_gelu_residual_f16x3" --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/p2 -- python tools/probe/f16x3_bench.py 45056 > $O/p2.log 2>&1
python - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/r06l/p1", "gpurun_out/r06l/p2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[(r["Kernel_Name"][:60], r["Counter_Name"])][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for (k, c), v in sorted(acc.items()):
        vals = list(v.values())
        print("%-62s %-28s mean %.4g  (n %d)" % (k, c, sum(vals) / len(vals), len(vals)))
PY
