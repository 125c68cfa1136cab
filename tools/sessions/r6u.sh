# round 6, final tree: counter passes, kernel traces and bench lines of every task / policy regime (tools/profile_r06.sh), BASELINE config 5 with its BESO policy
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/r06p; mkdir -p gpurun_out/r06p
python -c "import subprocess; print(subprocess.run(['hipcc','--version'],capture_output=True,text=True).stdout.splitlines()[0])" > gpurun_out/r06p/hipcc_version.txt
FORCE=1 bash tools/profile_r06.sh avoiding:random: pushing:mlp: pushing:scripted_push:--steps=100 sorting:mlp: sorting:scripted_push:--steps=60 sorting:ddpm: inserting:scripted_push:--steps=60,--warmup=5,--preroll=300 stacking:scripted_stack:--steps=100,--warmup=5 aligning:scripted_align:--steps=200,--warmup=5 stacking:beso:--steps=40,--warmup=5 > gpurun_out/r06p/profile.log 2>&1
tail -15 gpurun_out/r06p/profile.log
