# round 6: where the Aligning sub-step goes (stats build, workgroup 0, the timed region of the bench command)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06ab; mkdir -p $O
D3IL_STATS_LIB=1 python bench.py --task aligning --policy scripted_align --steps 100 --warmup 5 --no-cpu-baseline 2>$O/aligning_stats.err | tail -1 > $O/bench_line_aligning_stats_build.json
grep "device stats" $O/aligning_stats.err
D3IL_STATS_LIB=1 python bench.py --task stacking --policy scripted_stack --steps 100 --warmup 5 --no-cpu-baseline 2>$O/stacking_stats.err | tail -1 > $O/bench_line_stacking_stats_build.json
grep "device stats" $O/stacking_stats.err
