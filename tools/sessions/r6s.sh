cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06s; mkdir -p $O
python tools/gpu_beso_profile.py 2>&1 | grep -v amdgpu.ids | cut -c1-200 > $O/beso_profile_final.log; grep -E "predict_batch|Self CUDA time total" $O/beso_profile_final.log; grep -E "k_mlp|k_linear|k_attention|k_layernorm|elementwise|Cijk|Cat|index" $O/beso_profile_final.log | head -20 | cut -c1-40,150-200
