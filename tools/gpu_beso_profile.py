"""Where the time of the BESO policy of BASELINE config 5 goes (4096 lanes, window 5, 16 sampling steps): top kernels by device time.
usage (GPU box): python tools/gpu_beso_profile.py"""
import os
import sys
import time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda:0")
pol = bench._random_beso(dev)
n = 4096
obs = torch.randn(n, 20, device=dev)
for _ in range(7):
    pol.predict_batch(obs)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    pol.predict_batch(obs)
torch.cuda.synchronize()
print("predict_batch: %.2f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    pol.predict_batch(obs)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=18, max_name_column_width=70))
