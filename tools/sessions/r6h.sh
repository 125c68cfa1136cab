cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06h; mkdir -p $O
for a in 0 1 2 4 3 7; do echo "ablate $a (1 no GELU, 2 no second product, 4 no weight stream)"; python tools/probe/f16x3_bench.py 45056 -DHX_ABLATE=$a 2>&1 | grep "mlp"; done | tee $O/ablate.log
