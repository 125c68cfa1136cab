#!/bin/bash
# round 5 session pg: the DDPM policy's predict chain as one captured graph per sub-batch - does Sorting with its own policy stop being host bound at S = 4?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5pg; mkdir -p $O
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/graph_equal.log
import torch, sys
sys.path.insert(0, ".")
import bench
dev = torch.device("cuda:0")
fixed = torch.randn(8, 1024, 2, device=dev)
class Noise:
    def __init__(self): self.k = 0
    def __call__(self, shape):
        return fixed[0][:shape[0]]
outs = []
for g in (False, True):
    pol = bench._random_ddpm(22, dev, graph=g); pol.noise_fn = Noise()
    torch.manual_seed(1); obs = torch.randn(1024, 22, device=dev, dtype=torch.float64)
    o = [pol.predict_batch(obs + 0.01 * k).clone() for k in range(3)]
    outs.append(torch.stack(o))
print("eager vs captured DDPM chain, fixed noise: max |difference| %.3e, outputs differ between calls: %s" % ((outs[0] - outs[1]).abs().max().item(), bool((outs[1][0] != outs[1][1]).any())))
PY
for G in 0 1; do for S in 1 4; do
  python bench.py --task sorting --policy ddpm --no-cpu-baseline --sub-batches $S --policy-graph $G 2>$O/err_g${G}_sb$S.log | tail -1 > $O/sorting_ddpm_g${G}_sb$S.json
done; done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]; print("%-36s %9d env-steps/s  ms %.3f  kernel %.3f" % (f.split("/")[-1][:-5], d["value"], d["ms_per_step"], r["kernel_ms"]))
    except Exception as e: print(f, "ERR", str(e)[:80]); print(open(f.replace("sorting_ddpm_","err_").replace(".json",".log")).read()[-800:])
PY
