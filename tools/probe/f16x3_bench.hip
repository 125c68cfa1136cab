// Stand-alone build of the split-f16 policy kernels (d3il_amd/csrc/policy_f16x3.h) with several workgroup shapes, for tools/probe/f16x3_bench.py (seconds to compile).
#include <hip/hip_runtime.h>
#include "../../d3il_amd/csrc/policy_f16x3.h"
using namespace d3il;
extern "C" {
#define MLP(NW) int mlp_##NW(const float* h, const float* lw, const float* lb, float eps, const float* x, const void* wp, const float* b1, const float* b2, float* out, long rows, void* st) { \
  hipLaunchKernelGGL(k_mlp_gelu_residual_f16x3<NW>, dim3((unsigned)((rows + 16 * NW - 1) / (16 * NW))), dim3(64 * NW), 0, (hipStream_t)st, h, x, (const hx_h8*)wp, b1, b2, out, rows, lw, lb, eps); return (int)hipGetLastError(); }
#define LIN(NW) int lin_##NW(const float* xin, const float* lw, const float* lb, float eps, const void* wp, const float* bias, const float* resid, float* out, long rows, int N, void* st) { \
  hipLaunchKernelGGL(k_linear120_f16x3<NW>, dim3((unsigned)((rows + 16 * NW - 1) / (16 * NW))), dim3(64 * NW), 0, (hipStream_t)st, xin, (const hx_h8*)wp, bias, resid, out, rows, N, lw, lb, eps); return (int)hipGetLastError(); }
MLP(4) MLP(8) LIN(4) LIN(8) LIN(16)
}
