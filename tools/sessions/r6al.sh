# round 6: counter passes, kernel traces and bench lines of the two cooperative-engine tasks once more on the final library
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/r06p; mkdir -p gpurun_out/r06p
FORCE=1 bash tools/profile_r06.sh stacking:scripted_stack:--steps=100,--warmup=5 aligning:scripted_align:--steps=200,--warmup=5 > gpurun_out/r06p/profile.log 2>&1
tail -4 gpurun_out/r06p/profile.log
python bench.py --task aligning --policy mlp --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r06p/bench_line_aligning_mlp.json
