// policy_f16x3.h - the two GEMM kernels of the batched DiffusionGPT block (agents/models/beso/agents/diffusion_agents/k_diffusion/score_gpts.py:83-115)
// on the f16 matrix cores with SPLIT operands (included by rollout.hip).
//
// BASELINE config 5 (Stacking, 4096 environments per GPU, BESO policy) is policy bound: 1.5 TFLOP of f32 GEMM work per policy step (45056 token rows x 6 blocks x
// 16 sampling steps) is 9.5 ms at the f32 matrix peak of the chip (157 TFLOP/s; round 5 reached half of it).  v_mfma_f32_16x16x32_f16 runs at 16 x that rate, so an
// f32 product can be paid for with THREE f16 products and still be five times cheaper:
//     x = xh + 2^-11 xl,   xh = f16(x) (11 significant bits),  xl = f16((x - xh) 2^11) (the next 11 bits; the subtraction is exact in f32),
//     w x = wh xh + 2^-11 (wh xl + wl xh) + 2^-22 wl xl;
// the last term (2^-22 relative, below the f32 rounding of the sum it belongs to) is dropped.  The products are exact in the f32 accumulators of the matrix
// core (22-bit significands); `hh` and the cross terms accumulate separately and are combined once per output (acc_hh + 2^-11 acc_x).  Operands carry 22 of the 24
// bits of an f32, so a 120-term dot product differs from the f32-FMA chain by about as much as two f32 summation orders differ from each other (measured against an
// f64 reference in tests/test_policies_f16x3.py).  Range: |operand| <= 65504 (saturating) - LayerNorm outputs, GELU outputs and weights are O(1).
// The kernels keep the structure of k_mlp_gelu_residual_f32 / k_linear120_f32 (rollout.hip): one wave owns 16 token rows, the transposed product D[feature][row]
// makes the D registers of the first MLP product the B operand of the second, four waves share the weight stream through a double-buffered LDS stage.
#pragma once

namespace d3il {

typedef _Float16 hx_h8 __attribute__((ext_vector_type(8)));
typedef float hx_f4 __attribute__((ext_vector_type(4)));
constexpr int HX_C = 120, HX_H = 480, HX_PAIRS = HX_H / 32;      // a "pair" = two 16-wide hidden tiles = one K = 32 step of the second product
constexpr int HX_PAIR_V = 2048;                                  // 16-byte vectors per packed pair: 1024 first product (tile, step, half, lane) + 1024 second (out tile, half, lane)
constexpr float HX_LO = 2048.f, HX_ILO = 1.f / 2048.f;
#ifndef D3IL_HX_MLP_NW
#define D3IL_HX_MLP_NW 4
#endif
#ifndef D3IL_HX_LIN_NW
#define D3IL_HX_LIN_NW 4
#endif
constexpr int HX_MLP_NW = D3IL_HX_MLP_NW, HX_LIN_NW = D3IL_HX_LIN_NW;      // waves (of 16 rows) per workgroup of the two kernels

__device__ __forceinline__ void hx_split(float x, _Float16& hi, _Float16& lo) {
  x = fminf(fmaxf(x, -65504.f), 65504.f);
  hi = (_Float16)x;
  lo = (_Float16)((x - (float)hi) * HX_LO);
}
// Row `rr` of src[.][120] in the B-operand order of v_mfma_f32_16x16x32_f16: lane (g, j) holds, for K step s, the elements 32 s + 8 g + e (e = 0 .. 7) of row j
// (zero beyond 120), optionally layer-normalised first (nn.LayerNorm over the 120 features, biased variance: the four lane groups of a row hold 32 / 32 / 32 / 24 of
// its elements, the two row sums take two cross-group exchanges), split into the f16 halves.
__device__ __forceinline__ void hx_load_row(const float* __restrict__ src, long rr, int g, const float* __restrict__ ln_w, const float* __restrict__ ln_b, float eps,
                                            hx_h8* hh, hx_h8* hl) {
  float v[32];
#pragma unroll
  for (int s = 0; s < 4; s++) {
    const int k0 = 32 * s + 8 * g;
    if (k0 < HX_C) {
      const hx_f4 a = *(const hx_f4*)(src + rr * HX_C + k0), b = *(const hx_f4*)(src + rr * HX_C + k0 + 4);
#pragma unroll
      for (int e = 0; e < 4; e++) { v[8 * s + e] = a[e]; v[8 * s + 4 + e] = b[e]; }
    } else {
#pragma unroll
      for (int e = 0; e < 8; e++) v[8 * s + e] = 0.f;
    }
  }
  if (ln_w) {
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < 32; k++) sum += v[k];
    sum += __shfl_xor(sum, 16); sum += __shfl_xor(sum, 32);
    const float mean = sum * (1.0f / HX_C);
    float var = 0.f;
#pragma unroll
    for (int s = 0; s < 4; s++) if (32 * s + 8 * g < HX_C) {
#pragma unroll
      for (int e = 0; e < 8; e++) { const float d = v[8 * s + e] - mean; var += d * d; }
    }
    var += __shfl_xor(var, 16); var += __shfl_xor(var, 32);
    const float rstd = 1.0f / sqrtf(var * (1.0f / HX_C) + eps);
#pragma unroll
    for (int s = 0; s < 4; s++) {
      const int k0 = 32 * s + 8 * g;
      if (k0 < HX_C) {
#pragma unroll
        for (int e = 0; e < 8; e++) v[8 * s + e] = (v[8 * s + e] - mean) * rstd * ln_w[k0 + e] + ln_b[k0 + e];
      }
    }
  }
#pragma unroll
  for (int s = 0; s < 4; s++)
#pragma unroll
    for (int e = 0; e < 8; e++) { _Float16 a, b; hx_split(v[8 * s + e], a, b); hh[s][e] = a; hl[s][e] = b; }
}
// erf(x) without branches: libm's erff takes one of two paths per lane (|x| < 1: odd polynomial; else 1 - exp(-p(|x|))) and a wave with both kinds of lanes runs both
// under exec masks, with a jump each - eight evaluations per lane and hidden pair were the longest serial part of the MLP kernel.  The same two minimax forms
// (coefficients of the device library's erff), both evaluated by every lane, one select.
__device__ __forceinline__ float hx_erf(float x) {
  const float ax = fabsf(x), t = x * x;
  float p = fmaf(t, -0x1.268bc2p-11f, 0x1.420828p-8f);
  p = fmaf(t, p, -0x1.b5937p-6f); p = fmaf(t, p, 0x1.ce077cp-4f); p = fmaf(t, p, -0x1.81266p-2f); p = fmaf(t, p, 0x1.06ebap-3f);
  const float small = fmaf(ax, p, ax);
  float q = fmaf(ax, 0x1.1d3156p-16f, -0x1.8d129p-12f);
  q = fmaf(ax, q, 0x1.f9a6d2p-9f); q = fmaf(ax, q, -0x1.8c3164p-6f); q = fmaf(ax, q, 0x1.b4e9c8p-4f); q = fmaf(ax, q, 0x1.4515fap-1f); q = fmaf(ax, q, 0x1.078e5p-3f);
  q = fmaf(ax, q, ax);
  // exp(-q) = 2^(-q log2 e), the product in two pieces so that the fraction handed to v_exp_f32 keeps its low bits
  const float e_hi = q * -0x1.715476p+0f, e_lo = fmaf(q, -0x1.715476p+0f, -e_hi) + q * -0x1.4ae0bep-26f;
  const float n = rintf(e_hi);
  const float large = 1.0f - ldexpf(__builtin_amdgcn_exp2f((e_hi - n) + e_lo), (int)n);
  return copysignf(ax < 1.0f ? small : large, x);
}
// the same on two values at once: packed f32 arithmetic (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) halves the instruction count of the two polynomials - the
// MLP kernel is VALU bound in its GELU (profiles/r06/f16x3_mlp_ablation.log)
typedef float hx_f2 __attribute__((ext_vector_type(2)));
typedef _Float16 hx_h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ hx_f2 hx_erf2(hx_f2 x) {
  const hx_f2 ax = __builtin_elementwise_abs(x), t = x * x;
  auto F = [](hx_f2 a, hx_f2 b, float c) { return __builtin_elementwise_fma(a, b, hx_f2{c, c}); };
  hx_f2 p = F(t, hx_f2{-0x1.268bc2p-11f, -0x1.268bc2p-11f}, 0x1.420828p-8f);
  p = F(t, p, -0x1.b5937p-6f); p = F(t, p, 0x1.ce077cp-4f); p = F(t, p, -0x1.81266p-2f); p = F(t, p, 0x1.06ebap-3f);
  const hx_f2 small = __builtin_elementwise_fma(ax, p, ax);
  hx_f2 q = F(ax, hx_f2{0x1.1d3156p-16f, 0x1.1d3156p-16f}, -0x1.8d129p-12f);
  q = F(ax, q, 0x1.f9a6d2p-9f); q = F(ax, q, -0x1.8c3164p-6f); q = F(ax, q, 0x1.b4e9c8p-4f); q = F(ax, q, 0x1.4515fap-1f); q = F(ax, q, 0x1.078e5p-3f);
  q = __builtin_elementwise_fma(ax, q, ax);
  // exp(-q) = 2^(-q log2 e) by ONE v_exp_f32 per value: the product's rounding (2^-24 |e|) costs 2^-24 |e| ln 2 2^-|e| of the result - below 2e-8 wherever this branch
  // is taken (|x| >= 1: e >= 2.6) - so the two-piece product and the rint / ldexp range reduction of libm's expf are not needed here
  const hx_f2 e = q * hx_f2{-0x1.715476p+0f, -0x1.715476p+0f};
  hx_f2 r;
#pragma unroll
  for (int k = 0; k < 2; k++) r[k] = copysignf(ax[k] < 1.0f ? small[k] : 1.0f - __builtin_amdgcn_exp2f(e[k]), x[k]);
  return r;
}
// two values at once into their f16 halves (one packed conversion each way)
__device__ __forceinline__ void hx_split2(hx_f2 v, hx_h2& hi, hx_h2& lo) {
  v[0] = __builtin_amdgcn_fmed3f(v[0], -65504.f, 65504.f); v[1] = __builtin_amdgcn_fmed3f(v[1], -65504.f, 65504.f);
  hi = __builtin_convertvector(v, hx_h2);
  const hx_f2 back = __builtin_convertvector(hi, hx_f2);
  lo = __builtin_convertvector((v - back) * HX_LO, hx_h2);
}
#define HX_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)

// out[M][120] = x + b2 + W2 GELU(W1 LN(h) + b1).  wp: HX_STAGES = HX_PAIRS + 1 packed stages of 2048 vectors (policies.py pack_mlp_weights_f16x3):
//   stage k, vector ((tile * 4 + s) * 2 + p) * 64 + lane     = W1_p[32 k + 16 tile + i][32 s + 8 g + e]                     (first product of hidden pair k; zero for k = HX_PAIRS)
//   stage k, vector 1024 + (t * 2 + p) * 64 + lane           = W2_p[16 t + i][32 (k - 1) + 16 (e >> 2) + 4 g + (e & 3)]      (second product of hidden pair k - 1; zero for k = 0)
// with lane = 16 g + i, p = 0 the high half, p = 1 the low half times 2^11, zero beyond the matrices.  The second product sums the 32 hidden units of a pair in
// the order the two D tiles of the first product hold them: lane (g, j) has units 4 g + r of tile 0 in elements r and of tile 1 in elements 4 + r of its B operand.
// SOFTWARE PIPELINE: stage k issues the first product of pair k, the GELU of pair k - 1 and the second product of pair k - 1.  The three are independent inside a
// stage, so the matrix pipe works on one pair while the vector pipe computes the other's GELU (as one chain - first product, GELU, second product per pair - a
// wave's matrix and vector work did not overlap at all: profiles/r06/f16x3_mlp_ablation.log).  The two empty half stages (k = 0: zero W2, k = HX_PAIRS: zero W1)
// run like every other - no branches inside the loop body.
constexpr int HX_STAGES = HX_PAIRS + 1;
template <int NW>      // waves per workgroup: 16 NW rows share one pass over the weights
__global__ __launch_bounds__(64 * NW, 2) void k_mlp_gelu_residual_f16x3(const float* __restrict__ h, const float* __restrict__ x, const hx_h8* __restrict__ wp,
                                                                          const float* __restrict__ b1, const float* __restrict__ b2, float* __restrict__ out, long M,
                                                                          const float* __restrict__ ln_w, const float* __restrict__ ln_b, float eps) {
  __shared__ hx_h8 sw[2][HX_PAIR_V];
  __shared__ hx_f4 sb1[HX_H / 4 + 8];      // fc1's bias, one zero row of 32 in front (the GELU of "pair -1" in stage 0 reads it)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, g = lane >> 4;
  constexpr int NT = 64 * NW, VPT = HX_PAIR_V / NT;      // threads, 16-byte vectors per thread and stage
  const long row = (long)blockIdx.x * (16 * NW) + wave * 16 + j;
  const bool live = row < M;
  const long rr = live ? row : (M - 1);
  if (tid < 8) sb1[tid] = hx_f4{0.f, 0.f, 0.f, 0.f};
  if (tid < HX_H / 4) sb1[8 + tid] = ((const hx_f4*)b1)[tid];
  hx_h8 hh[4], hl[4];
  hx_load_row(h, rr, g, ln_w, ln_b, eps, hh, hl);
  hx_f4 acc2h[8], acc2x[8];
#pragma unroll
  for (int t = 0; t < 8; t++) { acc2h[t] = hx_f4{0.f, 0.f, 0.f, 0.f}; acc2x[t] = hx_f4{0.f, 0.f, 0.f, 0.f}; }
  hx_f4 ch[2], cx[2];      // first-product accumulators of the PREVIOUS pair (hh and cross terms), consumed by this stage's GELU
#pragma unroll
  for (int tile = 0; tile < 2; tile++) { ch[tile] = hx_f4{0.f, 0.f, 0.f, 0.f}; cx[tile] = ch[tile]; }
  hx_h8 pre[VPT];
#pragma unroll
  for (int q = 0; q < VPT; q++) sw[0][tid + NT * q] = wp[tid + NT * q];
  __syncthreads();
  for (int k = 0; k < HX_STAGES; k++) {
    const int cur = k & 1;
#if defined(HX_ABLATE) && (HX_ABLATE & 4)
    if (false)
#else
    if (k + 1 < HX_STAGES)
#endif
    {
#pragma unroll
      for (int q = 0; q < VPT; q++) pre[q] = wp[(long)(k + 1) * HX_PAIR_V + tid + NT * q];
    }
    // ---- first product of pair k (results are used by the NEXT stage)
    hx_f4 nh[2], nx1[2], nx2[2];
#pragma unroll
    for (int tile = 0; tile < 2; tile++) {
      nh[tile] = hx_f4{0.f, 0.f, 0.f, 0.f}; nx1[tile] = nh[tile]; nx2[tile] = nh[tile];
#pragma unroll
      for (int s = 0; s < 4; s++) {
        const hx_h8 wh = sw[cur][((tile * 4 + s) * 2) * 64 + lane], wl = sw[cur][((tile * 4 + s) * 2 + 1) * 64 + lane];
        nh[tile] = HX_MFMA(wh, hh[s], nh[tile]);
        nx1[tile] = HX_MFMA(wh, hl[s], nx1[tile]);
        nx2[tile] = HX_MFMA(wl, hh[s], nx2[tile]);
      }
    }
    // ---- GELU of pair k - 1
    hx_h8 gh, gl;
#pragma unroll
    for (int tile = 0; tile < 2; tile++) {
      const hx_f4 bb1 = sb1[8 * k + 4 * tile + g];      // (row 8 (k - 1) + 8 of the padded table)
      const hx_f4 v4 = ch[tile] + cx[tile] * HX_ILO + bb1;
#pragma unroll
      for (int r = 0; r < 4; r += 2) {
        const hx_f2 v = hx_f2{v4[r], v4[r + 1]};
#if defined(HX_ABLATE) && (HX_ABLATE & 1)
        const hx_f2 gv = v;      // (probe builds only: tools/probe/f16x3_bench.py -DHX_ABLATE=bits)
#else
        const hx_f2 gv = (v * 0.5f) * (hx_erf2(v * 0.70710678118654752440f) + 1.0f);      // nn.GELU() (exact form)
#endif
        hx_h2 a2, b2;
        hx_split2(gv, a2, b2);
        gh[4 * tile + r] = a2[0]; gh[4 * tile + r + 1] = a2[1]; gl[4 * tile + r] = b2[0]; gl[4 * tile + r + 1] = b2[1];
      }
    }
    // ---- second product of pair k - 1
#if defined(HX_ABLATE) && (HX_ABLATE & 2)
    acc2h[0][0] += (float)gh[0] + (float)gl[0] + (float)gh[5] + (float)gl[6];
#else
#pragma unroll
    for (int t = 0; t < 8; t++) acc2h[t] = HX_MFMA(sw[cur][1024 + (t * 2) * 64 + lane], gh, acc2h[t]);
#pragma unroll
    for (int t = 0; t < 8; t++) acc2x[t] = HX_MFMA(sw[cur][1024 + (t * 2) * 64 + lane], gl, acc2x[t]);
#pragma unroll
    for (int t = 0; t < 8; t++) acc2x[t] = HX_MFMA(sw[cur][1024 + (t * 2 + 1) * 64 + lane], gh, acc2x[t]);
#endif
#pragma unroll
    for (int tile = 0; tile < 2; tile++) { ch[tile] = nh[tile]; cx[tile] = nx1[tile] + nx2[tile]; }
#if defined(HX_ABLATE) && (HX_ABLATE & 4)
    if (false)
#else
    if (k + 1 < HX_STAGES)
#endif
    {
#pragma unroll
      for (int q = 0; q < VPT; q++) sw[cur ^ 1][tid + NT * q] = pre[q];
    }
    __syncthreads();
  }
  if (!live) return;
#pragma unroll
  for (int t = 0; t < 8; t++) {
    const int col = 16 * t + 4 * g;
    if (col >= HX_C) continue;
    const hx_f4 xr = *(const hx_f4*)(x + row * HX_C + col), bb = *(const hx_f4*)(b2 + col);
    *(hx_f4*)(out + row * HX_C + col) = xr + bb + acc2h[t] + acc2x[t] * HX_ILO;
  }
}

// out[M][N] = (LayerNorm)(xin)[M][120] W^T + bias (+ resid).  wp: ceil(N / 16) tiles of 512 vectors (policies.py pack_linear120_weights_f16x3):
//   vector (s * 2 + p) * 64 + lane of tile t = W_p[16 t + i][32 s + 8 g + e], zero beyond N rows / 120 columns.  Two tiles per LDS stage.
template <int NW>
__global__ __launch_bounds__(64 * NW) void k_linear120_f16x3(const float* __restrict__ xin, const hx_h8* __restrict__ wp, const float* __restrict__ bias, const float* __restrict__ resid,
                                                          float* __restrict__ out, long M, int N, const float* __restrict__ ln_w, const float* __restrict__ ln_b, float eps) {
  __shared__ hx_h8 sw[2][1024];
  __shared__ hx_f4 sbias[96];      // N <= 384 (checked by the caller)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, g = lane >> 4;
  constexpr int NT = 64 * NW, VPT = 1024 / NT;
  const long row = (long)blockIdx.x * (16 * NW) + wave * 16 + j;
  const bool live = row < M;
  const long rr = live ? row : (M - 1);
  if (tid < 96) sbias[tid] = 4 * tid < N ? ((const hx_f4*)bias)[tid] : hx_f4{0.f, 0.f, 0.f, 0.f};
  hx_h8 hh[4], hl[4];
  hx_load_row(xin, rr, g, ln_w, ln_b, eps, hh, hl);
  const int ntiles = (N + 15) / 16, nstages = (ntiles + 1) / 2;      // (the packed buffer is padded to an even number of tiles)
  hx_h8 pre[VPT];
#pragma unroll
  for (int q = 0; q < VPT; q++) sw[0][tid + NT * q] = wp[tid + NT * q];
  __syncthreads();
  for (int st = 0; st < nstages; st++) {
    const int cur = st & 1;
    if (st + 1 < nstages) {
#pragma unroll
      for (int q = 0; q < VPT; q++) pre[q] = wp[(long)(st + 1) * 1024 + tid + NT * q];
    }
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int t = 2 * st + u;
      hx_f4 ah = hx_f4{0.f, 0.f, 0.f, 0.f}, ax1 = ah, ax2 = ah;
#pragma unroll
      for (int s = 0; s < 4; s++) {
        const hx_h8 wh = sw[cur][512 * u + (s * 2) * 64 + lane], wl = sw[cur][512 * u + (s * 2 + 1) * 64 + lane];
        ah = HX_MFMA(wh, hh[s], ah);
        ax1 = HX_MFMA(wh, hl[s], ax1);
        ax2 = HX_MFMA(wl, hh[s], ax2);
      }
      const int col = 16 * t + 4 * g;
      if (live && col < N) {
        hx_f4 v = ah + (ax1 + ax2) * HX_ILO + sbias[col >> 2];
        if (resid) v += *(const hx_f4*)(resid + row * (long)N + col);
        *(hx_f4*)(out + row * (long)N + col) = v;
      }
    }
    if (st + 1 < nstages) {
#pragma unroll
      for (int q = 0; q < VPT; q++) sw[cur ^ 1][tid + NT * q] = pre[q];
    }
    __syncthreads();
  }
}

// The attention half of a DiffusionGPT block as ONE kernel: x1 = x + proj(causal_attention(ln1(x) Wqkv' + bqkv)) + bproj  (score_gpts.py:35-80, 103-107).
// As three kernels (ln1 + qkv product -> attention -> projection + residual) the half moved 65 MB of q | k | v out to HBM and back plus the attention output
// (21.6 MB each way) per block and sampling step at 45056 token rows - those kernels ran at 3 - 3.5 TB/s, i.e. memory bound (profiles/r06/beso_profile_final.log).
// Here ONE WAVE OWNS ONE SEQUENCE (T <= 16 tokens = the 16 rows of a matrix-core tile; T = 11 for the BESO policy: 5 of 16 rows are padding, which costs matrix
// work the half has to spare): the q | k | v rows of the sequence go from the D registers of the first product into LDS, the attention reads them there with the
// arithmetic of k_attention_causal_f32 in the same order (online softmax, key 0 peeled; lane (g, j) serves token j and the heads g and g + 4), writes its output over
// the q part of the row, and the projection takes its B operand from those rows.  HBM sees x once in and x1 once out.
// wp: 32 packed tiles (16 LDS stages of two): the 24 tiles of pack_linear120_weights_f16x3(cat(Wq, Wk, Wv)) followed by the 8 tiles of the projection.
// Two instantiations: <8 sequences, 11 rows> for T <= 11 (the BESO policy: 145 KB of LDS = one workgroup of eight waves per CU, two waves per SIMD) and <4, 16> for
// T <= 16.  The weight stream: two tiles (16 KB) per stage, double buffered.  At 45056 token rows: 59.9 us against 69.5 us for the three kernels it replaces
// (87.9 us with four waves per workgroup and one lane per (token, head pair); 65.4 us as two workgroups of four waves per CU with a single weight buffer and
// two barriers per tile - profiles/r06/attn_half/).
template <int NW, int TR>
__global__ __launch_bounds__(64 * NW) void k_attn_half_f16x3(const float* __restrict__ x, const hx_h8* __restrict__ wp, const float* __restrict__ bqkv, const float* __restrict__ bproj,
                                                             float* __restrict__ out, long B, int T, const float* __restrict__ ln_w, const float* __restrict__ ln_b, float eps) {
  constexpr int C = HX_C, H = 6, D = 20, QKV_STAGES = 12, STAGES = 16, RS = 3 * HX_C;
  __shared__ hx_h8 sw[2][1024];
  __shared__ hx_f4 sbias[96 + 32];      // q | k | v bias (90 vectors, zero padded to 96), then the projection's (30, padded to 32)
  __shared__ float qs[NW][TR][RS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, g = lane >> 4;
  constexpr int NT = 64 * NW, VPT = 1024 / NT;
  const long seq = (long)blockIdx.x * NW + wave;
  const bool live = seq < B && j < T;
  const long row = (seq < B ? seq : B - 1) * T + (j < T ? j : T - 1);
  for (int q = tid; q < 128; q += NT) sbias[q] = q < 90 ? ((const hx_f4*)bqkv)[q] : (q >= 96 && q < 126 ? ((const hx_f4*)bproj)[q - 96] : hx_f4{0.f, 0.f, 0.f, 0.f});
  hx_h8 hh[4], hl[4];
  hx_load_row(x, row, g, ln_w, ln_b, eps, hh, hl);
  hx_h8 pre[VPT];
#pragma unroll
  for (int q = 0; q < VPT; q++) sw[0][tid + NT * q] = wp[tid + NT * q];
  __syncthreads();
  float (*const my)[RS] = qs[wave];
  for (int st = 0; st < STAGES; st++) {
    const int cur = st & 1;
    if (st + 1 < STAGES) {
#pragma unroll
      for (int q = 0; q < VPT; q++) pre[q] = wp[(long)(st + 1) * 1024 + tid + NT * q];
    }
    if (st == QKV_STAGES) {
      // ---- causal attention of this wave's sequence on the rows in LDS (every lane of the workgroup has passed the barrier of stage 11: the rows are complete).
      // Lane (head h, pair p) serves the queries p and T - 1 - p: T + 1 key steps whatever p is (as k_attention_causal_f32)
      const float scale = 1.0f / sqrtf((float)D);
      const int TP = (T + 1) / 2, h = lane / TP, p = lane - h * TP;
      if (h < H) {
        auto ld = [&](const float* ptr, float* r) {
#pragma unroll
          for (int c = 0; c < D / 4; c++) { const hx_f4 t4 = *(const hx_f4*)(ptr + 4 * c); r[4 * c] = t4[0]; r[4 * c + 1] = t4[1]; r[4 * c + 2] = t4[2]; r[4 * c + 3] = t4[3]; }
        };
        for (int half = 0; half < 2; half++) {
          const int i = half == 0 ? p : T - 1 - p;
          if (half == 1 && i == p) break;      // the middle query of an odd T
          float q[D], acc[D], kk[D], vv[D];
          ld(&my[i][h * D], q);
#pragma unroll
          for (int d = 0; d < D; d++) q[d] *= scale;
          ld(&my[0][C + h * D], kk); ld(&my[0][2 * C + h * D], acc);
          float m = 0.0f, l = 1.0f;
#pragma unroll
          for (int d = 0; d < D; d++) m += q[d] * kk[d];
          for (int jj = 1; jj <= i; jj++) {
            ld(&my[jj][C + h * D], kk); ld(&my[jj][2 * C + h * D], vv);
            float sc = 0.0f;
#pragma unroll
            for (int d = 0; d < D; d++) sc += q[d] * kk[d];
            const float mn = fmaxf(m, sc), corr = __expf(m - mn), pj = __expf(sc - mn);
            l = l * corr + pj;
#pragma unroll
            for (int d = 0; d < D; d++) acc[d] = acc[d] * corr + pj * vv[d];
            m = mn;
          }
          const float inv = 1.0f / l;
#pragma unroll
          for (int c = 0; c < D / 4; c++) *(hx_f4*)(&my[i][h * D + 4 * c]) = hx_f4{acc[4 * c] * inv, acc[4 * c + 1] * inv, acc[4 * c + 2] * inv, acc[4 * c + 3] * inv};
        }
      }
      __syncthreads();
      // ---- the attention output of token j as the B operand of the projection
      const int jr = j < T ? j : T - 1;
#pragma unroll
      for (int s = 0; s < 4; s++) {
        const int k0 = 32 * s + 8 * g;
#pragma unroll
        for (int e = 0; e < 8; e++) {
          const float v = k0 < C ? my[jr][k0 + e] : 0.f;
          _Float16 a, b; hx_split(v, a, b); hh[s][e] = a; hl[s][e] = b;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 2; u++) {
      hx_f4 ah = hx_f4{0.f, 0.f, 0.f, 0.f}, ax1 = ah, ax2 = ah;
#pragma unroll
      for (int s = 0; s < 4; s++) {
        const hx_h8 wh = sw[cur][512 * u + (s * 2) * 64 + lane], wl = sw[cur][512 * u + (s * 2 + 1) * 64 + lane];
        ah = HX_MFMA(wh, hh[s], ah);
        ax1 = HX_MFMA(wh, hl[s], ax1);
        ax2 = HX_MFMA(wl, hh[s], ax2);
      }
      if (st < QKV_STAGES) {
        const int col = 16 * (2 * st + u) + 4 * g;
        if (j < T && col < 3 * C) *(hx_f4*)(&my[j][col]) = ah + (ax1 + ax2) * HX_ILO + sbias[col >> 2];
      } else {
        const int col = 16 * (2 * (st - QKV_STAGES) + u) + 4 * g;
        if (live && col < C) *(hx_f4*)(out + row * C + col) = ah + (ax1 + ax2) * HX_ILO + sbias[96 + (col >> 2)] + *(const hx_f4*)(x + row * C + col);
      }
    }
    if (st + 1 < STAGES) {
#pragma unroll
      for (int q = 0; q < VPT; q++) sw[cur ^ 1][tid + NT * q] = pre[q];
    }
    __syncthreads();
  }
}

}  // namespace d3il
