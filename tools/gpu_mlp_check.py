import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from d3il_amd import capi, policies as P
dev = torch.device("cuda:0")
torch.manual_seed(0)
C, H = 120, 480
fc1, fc2 = torch.nn.Linear(C, H).to(dev), torch.nn.Linear(H, C).to(dev)
L = capi.load()
for M in (1, 17, 64, 100, 4096 * 11):
    h, x = torch.randn(M, C, device=dev), torch.randn(M, C, device=dev)
    with torch.no_grad():
        ref = x + fc2(torch.nn.functional.gelu(fc1(h)))
        ref64 = (x.double() + torch.nn.functional.linear(torch.nn.functional.gelu(torch.nn.functional.linear(h.double(), fc1.weight.double(), fc1.bias.double())), fc2.weight.double(), fc2.bias.double()))
        wp = P.pack_mlp_weights(fc1, fc2)
        out = torch.empty_like(x)
        capi.check(L.d3il_mlp_gelu_residual_f32(h.data_ptr(), x.data_ptr(), wp.data_ptr(), fc1.bias.data_ptr(), fc2.bias.data_ptr(), out.data_ptr(), M, C, H, torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
    print("M %6d: |fused - torch f32| %.2e   |fused - f64| %.2e   |torch f32 - f64| %.2e" % (M, float((out - ref).abs().max()), float((out.double() - ref64).abs().max()), float((ref.double() - ref64).abs().max())))
M = 4096 * 11
h, x = torch.randn(M, C, device=dev), torch.randn(M, C, device=dev)
out = torch.empty_like(x)
with torch.no_grad():
    for name, fn in (("torch", lambda: x + fc2(torch.nn.functional.gelu(fc1(h)))),
                     ("fused (incl. weight packing)", lambda: capi.check(L.d3il_mlp_gelu_residual_f32(h.data_ptr(), x.data_ptr(), P.pack_mlp_weights(fc1, fc2).data_ptr(), fc1.bias.data_ptr(), fc2.bias.data_ptr(), out.data_ptr(), M, C, H, torch.cuda.current_stream().cuda_stream))),
                     ("fused kernel only", None)):
        if fn is None:
            wp = P.pack_mlp_weights(fc1, fc2)
            fn = lambda: capi.check(L.d3il_mlp_gelu_residual_f32(h.data_ptr(), x.data_ptr(), wp.data_ptr(), fc1.bias.data_ptr(), fc2.bias.data_ptr(), out.data_ptr(), M, C, H, torch.cuda.current_stream().cuda_stream))
        for _ in range(5): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
        print("%-30s %.1f us  (%.1f TFLOP/s)" % (name, dt * 1e6, 2 * M * 2 * C * H / dt / 1e12))
