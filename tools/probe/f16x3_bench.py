"""Kernel times of the split-f16 policy kernels by workgroup shape (GPU box): python tools/probe/f16x3_bench.py [rows]"""
import ctypes as C, os, subprocess, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from d3il_amd.policies import pack_linear120_weights_f16x3, pack_mlp_weights_f16x3
so = "/tmp/f16x3_bench%s.so" % "".join(a for a in sys.argv[2:] if a.startswith("-D")).replace("=", "_")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", so, os.path.join(ROOT, "tools", "probe", "f16x3_bench.hip")] +
                      [a for a in sys.argv[2:] if a.startswith("-D")])
L = C.CDLL(so)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 45056
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.randn(rows, 120, device=dev); out = torch.empty_like(x); out3 = torch.empty(rows, 360, device=dev)
W1, W2, Wq, Wp = torch.randn(480, 120, device=dev) * .05, torch.randn(120, 480, device=dev) * .05, torch.randn(360, 120, device=dev) * .05, torch.randn(120, 120, device=dev) * .05
b1, b2, bq = torch.zeros(480, device=dev), torch.zeros(120, device=dev), torch.zeros(360, device=dev)
lw, lb = torch.ones(120, device=dev), torch.zeros(120, device=dev)
wm, wq, wp = pack_mlp_weights_f16x3(W1, W2), pack_linear120_weights_f16x3(Wq), pack_linear120_weights_f16x3(Wp)
st = torch.cuda.current_stream(dev).cuda_stream
P = lambda t: C.c_void_p(t.data_ptr())
def timeit(f, n=30):
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for nw in (4, 8):
    f = getattr(L, "mlp_%d" % nw)
    us = timeit(lambda: f(P(x), P(lw), P(lb), C.c_float(1e-5), P(x), P(wm), P(b1), P(b2), P(out), C.c_long(rows), C.c_void_p(st)))
    print("mlp NW=%d: %.1f us  (%.0f TFLOP/s f32-equivalent, %.0f issued f16)" % (nw, us, 2 * rows * 57600 * 2 / us / 1e6, 3 * 2 * rows * 57600 * 2 / us / 1e6))
for nw in (4, 8, 16):
    f = getattr(L, "lin_%d" % nw)
    us = timeit(lambda: f(P(x), P(lw), P(lb), C.c_float(1e-5), P(wq), P(bq), None, P(out3), C.c_long(rows), 360, C.c_void_p(st)))
    us2 = timeit(lambda: f(P(x), None, None, C.c_float(0), P(wp), P(b2), P(x), P(out), C.c_long(rows), 120, C.c_void_p(st)))
    print("linear NW=%d: qkv %.1f us, proj %.1f us" % (nw, us, us2))
