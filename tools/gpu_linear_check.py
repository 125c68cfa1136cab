"""Time of d3il_linear120_f32 against torch (LayerNorm + Linear [+ residual]) for the two shapes of the DiffusionGPT block at 4096 x 11 rows."""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
from d3il_amd import capi, policies as P
dev = torch.device("cuda:0"); torch.manual_seed(0)
L = capi.load(); M = 4096 * 11
ln = torch.nn.LayerNorm(120).to(dev)
st = torch.cuda.current_stream().cuda_stream
for N, use_ln, use_res in ((360, True, False), (120, False, True)):
    lin = torch.nn.Linear(120, N).to(dev)
    x, res, out = torch.randn(M, 120, device=dev), torch.randn(M, N, device=dev), torch.empty(M, N, device=dev)
    wp = P.pack_linear120_weights(lin.weight)
    with torch.no_grad():
        fns = (("torch", lambda: lin(ln(x) if use_ln else x) + (res if use_res else 0)),
               ("fused", lambda: capi.check(L.d3il_linear120_f32(x.data_ptr(), ln.weight.data_ptr() if use_ln else None, ln.bias.data_ptr() if use_ln else None, 1e-5, wp.data_ptr(), lin.bias.data_ptr(),
                                                                  res.data_ptr() if use_res else None, out.data_ptr(), M, N, st))))
        for name, fn in fns:
            for _ in range(5): fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(50): fn()
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
            print("N %3d ln %d res %d %-6s %.1f us (%.1f TFLOP/s)" % (N, use_ln, use_res, name, dt * 1e6, 2 * M * 120 * N / dt / 1e12))
        ref = lin(ln(x) if use_ln else x) + (res if use_res else 0)
        print("   max |fused - torch| %.2e" % float((out - ref).abs().max()))
