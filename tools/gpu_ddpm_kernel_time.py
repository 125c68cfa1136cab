"""DDPM policy (BASELINE config 4's denoiser): time of one predict call alone on the GPU - the fused matrix-core chain (d3il_ddpm_mlp_f32) against the torch chain,
at the row counts of a sub-batch and of a whole batch.  usage (GPU box): python tools/gpu_ddpm_kernel_time.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from d3il_amd.policies import DDPMPolicy, DiffusionMLP, Scaler  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
net = DiffusionMLP(action_dim=2, obs_dim=16, t_dim=8, hidden_dim=256, num_hidden_layers=8).to(dev)
sc = Scaler([0.0] * 16, [1.0] * 16, [0.0, 0.0], [0.01, 0.01], y_bounds=[[-1.0, -1.0], [1.0, 1.0]], device=dev)
pol = DDPMPolicy(net, sc, n_timesteps=4, window_size=1)
flop_row = 4 * 2 * (28 * 256 + 8 * 256 * 256 + 256 * 16)      # issued on the matrix cores per row (padded first / last layer)
for n in (256, 1024, 4096, 16384):
    s = torch.randn(n, 16, device=dev)
    with torch.no_grad():
        for fused in (True, False):
            f = (lambda: pol._sample_fused(s)) if fused else (lambda: pol._sample(s))
            for _ in range(5):
                f()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); a.record()
            for _ in range(50):
                f()
            b.record(); torch.cuda.synchronize()
            ms = a.elapsed_time(b) / 50
            print("%6d rows, %s: %.3f ms per call%s" % (n, "fused kernel (+ one randn)" if fused else "torch chain", ms, ", %.1f TFLOP/s on the matrix cores" % (flop_row * n / ms / 1e9) if fused else ""))
