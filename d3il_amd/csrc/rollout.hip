// rollout.hip - HIP kernels (gfx950) and the C ABI of libd3il_rollout.so.
//
// One environment per lane, one wavefront (64 lanes) per workgroup: at N = 4096 that is 64 workgroups on 64
// different CUs, each wave alone on its SIMD (the path is FP64 VALU / latency bound, not HBM bound: the whole
// env step touches < 1 KB of HBM per environment).  State is SoA [field][env] so that every field is one
// coalesced 512-byte row per wave; the constant block is read with scalar loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <memory>
#include <mutex>
#include <string>
#include "../../include/d3il_rollout.h"
#include "panda_step.h"
#include "rigid_common.h"
#include "gen/avoiding_consts.inc"
#include "gen/stacking_consts.inc"

namespace d3il {
constexpr int WAVE = 64;
}
#include "gen_kernels.h"
#include "stack_kernels.h"
#include "policy_f16x3.h"

namespace d3il {

// The constant block is read through the constant address space so that every access is a scalar (SGPR) load.
typedef const __attribute__((address_space(4))) PandaConsts CPanda4;
__device__ __forceinline__ CPanda4* to_const_as(const PandaConsts* p) { return (CPanda4*)(unsigned long long)p; }

__device__ __forceinline__ void load_state(const double* __restrict__ state, const unsigned* __restrict__ flags,
                                           const int* __restrict__ steps, int stride, int e, EnvState& st) {
  const double* s = state + e;
#pragma unroll
  for (int i = 0; i < NDOF; i++) st.q[i] = s[(D3IL_STATE_QPOS + i) * (size_t)stride];
#pragma unroll
  for (int i = 0; i < NDOF; i++) st.v[i] = s[(D3IL_STATE_QVEL + i) * (size_t)stride];
#pragma unroll
  for (int i = 0; i < NARM; i++) st.bias[i] = s[(D3IL_STATE_BIAS + i) * (size_t)stride];
#pragma unroll
  for (int i = 0; i < 3; i++) st.tcp[i] = s[(D3IL_STATE_TCP + i) * (size_t)stride];
#pragma unroll
  for (int i = 0; i < NARM; i++) st.ikq[i] = s[(D3IL_STATE_IK_Q + i) * (size_t)stride];
#pragma unroll
  for (int i = 0; i < NARM; i++) st.ikqd[i] = s[(D3IL_STATE_IK_QD + i) * (size_t)stride];
  st.flags = flags[e]; st.step = steps[e];
}
__device__ __forceinline__ void store_state(double* __restrict__ state, unsigned* __restrict__ flags, int* __restrict__ steps,
                                            int stride, int e, const EnvState& st) {
  double* s = state + e;
#pragma unroll
  for (int i = 0; i < NDOF; i++) s[(D3IL_STATE_QPOS + i) * (size_t)stride] = st.q[i];
#pragma unroll
  for (int i = 0; i < NDOF; i++) s[(D3IL_STATE_QVEL + i) * (size_t)stride] = st.v[i];
#pragma unroll
  for (int i = 0; i < NARM; i++) s[(D3IL_STATE_BIAS + i) * (size_t)stride] = st.bias[i];
#pragma unroll
  for (int i = 0; i < 3; i++) s[(D3IL_STATE_TCP + i) * (size_t)stride] = st.tcp[i];
#pragma unroll
  for (int i = 0; i < NARM; i++) s[(D3IL_STATE_IK_Q + i) * (size_t)stride] = st.ikq[i];
#pragma unroll
  for (int i = 0; i < NARM; i++) s[(D3IL_STATE_IK_QD + i) * (size_t)stride] = st.ikqd[i];
  flags[e] = st.flags; steps[e] = st.step;
}
__device__ __forceinline__ void store_outputs(const EnvState& st, int e, const float* o, unsigned char dn, float* __restrict__ obs,
                                              unsigned char* __restrict__ done, unsigned char* __restrict__ success, unsigned short* __restrict__ mode) {
  obs[2 * e] = o[0]; obs[2 * e + 1] = o[1];
  done[e] = dn; success[e] = (st.flags & F_SUCCESS) ? 1 : 0; mode[e] = (unsigned short)(st.flags & F_MODE_MASK);
}

// env.step() for the Avoiding task: controller + physics fused over all sub-steps, state stays in registers.
template <bool FAST, bool BAKED>
__global__ __launch_bounds__(WAVE) void k_avoiding_step(const PandaConsts* __restrict__ cp, double* __restrict__ state,
                                                        unsigned* __restrict__ flags, int* __restrict__ steps,
                                                        const double* __restrict__ actions, float* __restrict__ obs,
                                                        unsigned char* __restrict__ done, unsigned char* __restrict__ success,
                                                        unsigned short* __restrict__ mode, int n, int stride, int n_substeps, int max_steps, int lanes) {
  int e = blockIdx.x * lanes + threadIdx.x;
  if ((int)threadIdx.x >= lanes || e >= n) return;
  EnvState st;
  load_state(state, flags, steps, stride, e, st);
  double act[7];
#pragma unroll
  for (int k = 0; k < 7; k++) act[k] = actions[(size_t)e * 7 + k];
  const bool bad_action = sanitize_action(act, actions + (size_t)e * 7);
  float o[2]; unsigned char dn;
#if defined(D3IL_DEVICE_STATS)
  unsigned long long t0 = wall_clock64();
#endif
  if constexpr (BAKED) env_step<FAST>(kAvoidingConsts, st, act, o, &dn, n_substeps, max_steps);
  else env_step<FAST>(*to_const_as(cp), st, act, o, &dn, n_substeps, max_steps);
#if defined(D3IL_DEVICE_STATS)
  if (threadIdx.x == 0 && blockIdx.x < 4096) g_dev_wave[blockIdx.x][9] = wall_clock64() - t0;
#endif
  if (bad_action) st.flags |= F_SOLVER_FAIL | F_TERMINATED;
  store_state(state, flags, steps, stride, e, st);
  store_outputs(st, e, o, dn, obs, done, success, mode);
}

// env.step(), two cooperating waves per 64 environments.
//
// The reference's Cartesian controller integrates its virtual joint target open loop (joint_filter_coefficient = 1,
// IKControllers.py:171-176): the sequence of PD set-points of one env step depends only on the action
// and the controller state at the start of the step, not on the physics.  So the IK chain (60 % of a sub-step's
// instructions) and the physics chain (40 %) are two independent sequential pipelines.  Wave 0 of the workgroup runs the
// controller and publishes (q_des, qd_des) of sub-step s into an LDS slot; wave 1 runs forward dynamics, constraints
// and integration of sub-step s-1 meanwhile and picks the set-point up after one workgroup barrier per sub-step.
// The critical path per sub-step drops from IK + physics to max(IK, physics); at small N (4096 envs = 64 workgroups on
// 256 CUs) the extra wave runs on an otherwise idle SIMD.
//
// SERVE = true adds a third wave: the rare constraint paths of the physics (a rod contact; arm joint-limit rows) run THERE.  The physics wave
// posts, once per sub-step, the mask of its lanes that need one (operands through `rx`, RX_ROWS doubles per lane) and waits for the reply only
// when the mask is non-zero; the serving wave polls that word with s_sleep.  With the solvers inlined into the physics wave its hot path
// spilled 192 registers (68 scratch operations per sub-step in its main block) and a single environment in contact - about every second
// launch at 4096 environments - stretched its workgroup, hence the launch, from 0.40 to 0.9 ms; both go away.
struct RareXch {
  static constexpr bool remote = true;
  double* buf;     // LDS [RX_ROWS][WAVE]
  int* ctl;        // LDS: [0] sequence number posted by the physics wave (sub-step + 1), [1] acknowledged by the serving wave, [2..3] lane mask
  int lane, seq;
  __device__ __forceinline__ void put(int k, double v) { buf[k * WAVE + lane] = v; }
  __device__ __forceinline__ double get(int k) const { return buf[k * WAVE + lane]; }
  // physics wave (all lanes): true when some lane asked and the reply is in
  __device__ __forceinline__ bool post(bool need) {
    const unsigned long long m = __ballot(need);
    if (lane == 0) { ctl[2] = (int)(unsigned)m; ctl[3] = (int)(unsigned)(m >> 32); }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_store(&ctl[0], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (m == 0) return false;
    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&ctl[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) != seq) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return true;
  }
  // serving wave: the mask of sub-step `seq` (waits until it is posted)
  __device__ __forceinline__ unsigned long long wait_request() {
    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&ctl[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) != seq) __builtin_amdgcn_s_sleep(2);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return (unsigned long long)(unsigned)ctl[2] | ((unsigned long long)(unsigned)ctl[3] << 32);
  }
  __device__ __forceinline__ void acknowledge() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_store(&ctl[1], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
};
constexpr size_t AVOID_LDS_SERVE = (size_t)RX_ROWS * WAVE * sizeof(double) + 4 * sizeof(int);

template <bool FAST, bool SERVE>
__global__ __launch_bounds__((SERVE ? 3 : 2) * WAVE) void k_avoiding_step_split(const PandaConsts* __restrict__ cp, double* __restrict__ state,
                                                                  unsigned* __restrict__ flags, int* __restrict__ steps,
                                                                  const double* __restrict__ actions, float* __restrict__ obs,
                                                                  unsigned char* __restrict__ done, unsigned char* __restrict__ success,
                                                                  unsigned short* __restrict__ mode, int n, int stride, int n_substeps, int max_steps) {
  __shared__ double xch[2][2 * NARM][WAVE];
  __shared__ double trg[2][2 * NARM + 1][WAVE];     // sin / cos of ikq (controller) and of the arm joints (physics) carried across the sub-steps: parked in LDS between them
  extern __shared__ double rx_smem[];             // SERVE: the rare-path exchange area
  const int lane = threadIdx.x & (WAVE - 1);
  const int role = threadIdx.x / WAVE;          // wave-uniform: 0 controller, 1 physics, 2 (SERVE) the rare constraint paths
  RareXch rx; rx.buf = rx_smem; rx.ctl = (int*)(rx_smem + RX_ROWS * WAVE); rx.lane = lane; rx.seq = 0;
  int e = blockIdx.x * WAVE + lane;
  const bool live = e < n;
  if (!live) e = n - 1;                         // keep every lane in the barriers; dead lanes recompute env n-1 and store nothing
  const PandaConsts& c = kAvoidingConsts;
  (void)cp;
#if defined(D3IL_DEVICE_STATS)
  unsigned long long t0 = wall_clock64(), tw = 0;
#endif
  if (role == 0) {
    const double* sp = state + e;
    double ikq[NARM], ikqd[NARM], q0[NARM], act[7], des[7];
#pragma unroll
    for (int i = 0; i < NARM; i++) {
      ikq[i] = sp[(D3IL_STATE_IK_Q + i) * (size_t)stride]; ikqd[i] = sp[(D3IL_STATE_IK_QD + i) * (size_t)stride];
      q0[i] = sp[(D3IL_STATE_QPOS + i) * (size_t)stride];
    }
#pragma unroll
    for (int k = 0; k < 7; k++) act[k] = actions[(size_t)e * 7 + k];
    unsigned fl = flags[e];
    sanitize_action(act, actions + (size_t)e * 7);
    make_setpoint(act, des);
    double vwarm[7];
    vwarm[6] = 0.0; trg[0][2 * NARM][lane] = 0.0;
#pragma clang loop unroll(disable)
    for (int s = 0; s < n_substeps; s++) {
      {
        double trig[2 * NARM + 1];                 // exact once per step, then carried
#pragma unroll
        for (int k = 0; k <= 2 * NARM; k++) trig[k] = trg[0][k][lane];
        ik_update<FAST>(c, des, des + 3, q0, fl, ikq, ikqd, vwarm, trig);
#pragma unroll
        for (int k = 0; k <= 2 * NARM; k++) trg[0][k][lane] = trig[k];
      }
      const int b = s & 1;
#pragma unroll
      for (int k = 0; k < NARM; k++) { xch[b][k][lane] = ikq[k]; xch[b][NARM + k][lane] = ikqd[k]; }
#if defined(D3IL_DEVICE_STATS)
      unsigned long long tb = wall_clock64();
#endif
      __syncthreads();
#if defined(D3IL_DEVICE_STATS)
      tw += wall_clock64() - tb;
#endif
    }
#if defined(D3IL_DEVICE_STATS)
    if (lane == 0 && blockIdx.x < 4096) { g_dev_wave[blockIdx.x][8] = wall_clock64() - t0 - tw; }
#endif
    if (live) {
      double* so = state + e;
#pragma unroll
      for (int i = 0; i < NARM; i++) { so[(D3IL_STATE_IK_Q + i) * (size_t)stride] = ikq[i]; so[(D3IL_STATE_IK_QD + i) * (size_t)stride] = ikqd[i]; }
    }
  } else if (SERVE && role == 2) {
    double warm[6];
    warm[5] = 0.0;
    if (lane == 0) { rx.ctl[0] = 0; rx.ctl[1] = 0; }       // before the first barrier; the physics wave posts after it
#pragma clang loop unroll(disable)
    for (int s = 0; s < n_substeps; s++) {
      __syncthreads();
      rx.seq = s + 1;
      const unsigned long long m = rx.wait_request();
      if (m != 0) {                                        // wave-uniform
#if defined(D3IL_DEVICE_STATS)
        unsigned long long tb = wall_clock64();
#endif
        if ((m >> lane) & 1ull) rare_serve(c, &rx, warm);
        rx.acknowledge();
#if defined(D3IL_DEVICE_STATS)
        tw += wall_clock64() - tb;
#endif
      }
    }
#if defined(D3IL_DEVICE_STATS)
    if (lane == 0 && blockIdx.x < 4096) g_dev_wave[blockIdx.x][7] = tw;      // the serving wave's busy ticks (replaces the sub-step counter of this slot: written last)
#endif
  } else {
    EnvState st;
    load_state(state, flags, steps, stride, e, st);
    float o[2]; unsigned char dn;
    step_begin(c, st, o, &dn, max_steps);
    double warm[6];
    warm[5] = 0.0;
#pragma unroll
    for (int k = 0; k < NARM; k++) { double sk, ck; sincos(st.q[k], &sk, &ck); trg[1][k][lane] = sk; trg[1][NARM + k][lane] = ck; }
#pragma clang loop unroll(disable)
    for (int s = 0; s < n_substeps; s++) {
#if defined(D3IL_DEVICE_STATS)
      unsigned long long tb = wall_clock64();
#endif
      __syncthreads();
#if defined(D3IL_DEVICE_STATS)
      tw += wall_clock64() - tb;
#endif
      const int b = s & 1;
      double qd[NARM], qdd[NARM];
#pragma unroll
      for (int k = 0; k < NARM; k++) { qd[k] = xch[b][k][lane]; qdd[k] = xch[b][NARM + k][lane]; }
      double trig[2 * NARM];
#pragma unroll
      for (int k = 0; k < 2 * NARM; k++) trig[k] = trg[1][k][lane];
      if constexpr (SERVE) { rx.seq = s + 1; control_and_physics(c, st, qd, qdd, 0.04, false, warm, trig, &rx); }
      else control_and_physics(c, st, qd, qdd, 0.04, false, warm, trig);
#pragma unroll
      for (int k = 0; k < 2 * NARM; k++) trg[1][k][lane] = trig[k];
    }
#if defined(D3IL_DEVICE_STATS)
    if (lane == 0 && blockIdx.x < 4096) { g_dev_wave[blockIdx.x][9] = wall_clock64() - t0 - tw; }
#endif
    st.flags |= F_IK_VALID;                     // set by the controller wave's first ik_update in the fused kernel
    if (action_is_bad(actions + (size_t)e * 7)) st.flags |= F_SOLVER_FAIL | F_TERMINATED;
    step_end(c, st);
    if (live) {
      double* so = state + e;
#pragma unroll
      for (int i = 0; i < NDOF; i++) { so[(D3IL_STATE_QPOS + i) * (size_t)stride] = st.q[i]; so[(D3IL_STATE_QVEL + i) * (size_t)stride] = st.v[i]; }
#pragma unroll
      for (int i = 0; i < NARM; i++) so[(D3IL_STATE_BIAS + i) * (size_t)stride] = st.bias[i];
#pragma unroll
      for (int i = 0; i < 3; i++) so[(D3IL_STATE_TCP + i) * (size_t)stride] = st.tcp[i];
      flags[e] = st.flags; steps[e] = st.step;
      store_outputs(st, e, o, dn, obs, done, success, mode);
    }
  }
}

// env.reset() for masked environments
__global__ __launch_bounds__(WAVE) void k_avoiding_reset(const PandaConsts* __restrict__ cp, const double* __restrict__ init_qpos,
                                                         const unsigned char* __restrict__ mask, double* __restrict__ state,
                                                         unsigned* __restrict__ flags, int* __restrict__ steps, float* __restrict__ obs,
                                                         unsigned char* __restrict__ done, unsigned char* __restrict__ success,
                                                         unsigned short* __restrict__ mode, int n, int stride) {
  int e = blockIdx.x * WAVE + threadIdx.x;
  if (e >= n) return;
  if (mask && !mask[e]) return;
  EnvState st;
  double iq[NARM];
#pragma unroll
  for (int k = 0; k < NARM; k++) iq[k] = init_qpos[k];
  float o[2];
  env_reset(kAvoidingConsts, st, iq, o);
  store_state(state, flags, steps, stride, e, st);
  store_outputs(st, e, o, 0, obs, done, success, mode);
}

// Auto-reset of finished environments (the rollout loop's `env.reset()` at avoiding_sim.py:51 for the next trajectory),
// fused with the harness bookkeeping: episode counters += (finished, successful), desired pose := TCP after the reset
// (avoiding_sim.py:53-54).  One launch instead of reset + policy_begin + three reductions.
__global__ __launch_bounds__(WAVE) void k_avoiding_auto_reset(const PandaConsts* __restrict__ cp, const double* __restrict__ init_qpos,
                                                              double* __restrict__ state, unsigned* __restrict__ flags, int* __restrict__ steps,
                                                              float* __restrict__ obs, unsigned char* __restrict__ done,
                                                              unsigned char* __restrict__ success, unsigned short* __restrict__ mode,
                                                              double* __restrict__ des, long long* __restrict__ episode_counts, int n, int stride) {
  int e = blockIdx.x * WAVE + threadIdx.x;
  if (e >= n || !done[e]) return;
  (void)cp;
  atomicAdd((unsigned long long*)&episode_counts[0], 1ull);
  if (success[e]) atomicAdd((unsigned long long*)&episode_counts[1], 1ull);
  EnvState st;
  double iq[NARM];
#pragma unroll
  for (int k = 0; k < NARM; k++) iq[k] = init_qpos[k];
  float o[2];
  env_reset(kAvoidingConsts, st, iq, o);
  store_state(state, flags, steps, stride, e, st);
  store_outputs(st, e, o, 0, obs, done, success, mode);
#pragma unroll
  for (int k = 0; k < 3; k++) des[k * (size_t)stride + e] = st.tcp[k];
}

// Causal self-attention for the short token sequences of the BESO policy (DiffusionGPT of BASELINE config 5: T = 11 tokens, 6 heads
// of 20): one lane per (sequence b, head h, query i).  qkv: f32 [B * T][3 C] (query | key | value of a token, C = H * D, as written
// by ONE fused linear layer), out: f32 [B * T][C] in token-major layout (what the output projection reads) - no transposes, no
// [B, H, T, T] score tensor, no batched GEMM of 11 x 20 matrices.  Softmax over the keys j <= i with the 1 / sqrt(D) scaling
// (score_gpts.py:59-76).  D <= 32, T <= 32.
// LayerNorm over the last dimension for the narrow rows of the policy transformer (C = 120): 32 lanes per row, four consecutive
// floats per lane (C <= 128, C a multiple of 4), two rows per wave; mean and variance by xor shuffles inside the half wave; the
// biased variance and eps inside the square root like torch.nn.LayerNorm.  x, y: f32 [rows][C].
// Fused transformer MLP of the batched DiffusionGPT policy (score_gpts.py:83-115: x + fc2(GELU(fc1(ln2(x)))), n_embd 120, hidden 480), f32 on the matrix
// cores: out[M][120] = x + b2 + W2 GELU(W1 h + b1) for the [B * T][120] activations of the BESO policy.  One wave owns 16 rows; per chunk
// of 16 hidden units it runs the TRANSPOSED first product D1[hid][row] = W1c[16 x 120] h^T[120 x 16] with v_mfma_f32_16x16x4_f32 (30 steps), applies bias +
// GELU to its four D registers and feeds them STRAIGHT BACK as the B operand of the second product D2[out][row] += W2c[128 x 16] G[16 x 16] (8 tiles x 4
// steps) - the D layout (lane = row, register r of lane group g = hidden 4 g + r) is already a B layout when step e of the second product sums the hidden
// units {4 g + e}, which only fixes how the W2 chunk is packed - so the hidden activations never leave the registers.  Weights arrive pre-packed per chunk
// in exactly the LDS order (16 blocks of [4 g][16 i][4 e] floats, one float4 per lane: d3il_amd/policies.py pack_mlp_weights), double-buffered in LDS and shared by the four waves of
// a workgroup.  1860 MFMAs per 16 rows; 14.7 MFLOP per 64-row workgroup against 32 KB of weight traffic from L2.
typedef float mlp_f4 __attribute__((ext_vector_type(4)));
constexpr int MLP_C = 120, MLP_H = 480, MLP_CHUNKS = MLP_H / 16, MLP_CHUNK_F = 16 * 256;      // floats per packed chunk
// The row of a lane in the B-operand order of the products below: lane (g, j) holds elements 4 s + g of row j.  With ln_w the row is layer-normalised
// first (nn.LayerNorm over the 120 features, biased variance): the four lane groups of a row each hold 30 of its elements, so the two row sums are
// a 30-term sum per lane and two cross-group exchanges.
__device__ __forceinline__ void mlp_load_row(const float* __restrict__ src, long rr, int g, const float* __restrict__ ln_w, const float* __restrict__ ln_b, float eps, float* hk) {
#pragma unroll
  for (int s2 = 0; s2 < MLP_C / 4; s2++) hk[s2] = src[rr * MLP_C + 4 * s2 + g];
  if (ln_w) {
    float sum = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < MLP_C / 4; s2++) sum += hk[s2];
    sum += __shfl_xor(sum, 16); sum += __shfl_xor(sum, 32);
    const float mean = sum * (1.0f / MLP_C);
    float var = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < MLP_C / 4; s2++) { const float d = hk[s2] - mean; var += d * d; }
    var += __shfl_xor(var, 16); var += __shfl_xor(var, 32);
    const float rstd = 1.0f / sqrtf(var * (1.0f / MLP_C) + eps);
#pragma unroll
    for (int s2 = 0; s2 < MLP_C / 4; s2++) hk[s2] = (hk[s2] - mean) * rstd * ln_w[4 * s2 + g] + ln_b[4 * s2 + g];
  }
}
__global__ __launch_bounds__(256) void k_mlp_gelu_residual_f32(const float* __restrict__ h, const float* __restrict__ x, const float* __restrict__ wp,
                                                                const float* __restrict__ b1, const float* __restrict__ b2, float* __restrict__ out, long M,
                                                                const float* __restrict__ ln_w, const float* __restrict__ ln_b, float eps) {
  __shared__ mlp_f4 sw[2][MLP_CHUNK_F / 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, g = lane >> 4;
  const long row0 = (long)blockIdx.x * 64 + wave * 16, row = row0 + j;
  const bool live = row < M;
  const long rr = live ? row : (M - 1);
  float hk[MLP_C / 4];
  mlp_load_row(h, rr, g, ln_w, ln_b, eps, hk);
  mlp_f4 acc2[8];
#pragma unroll
  for (int t = 0; t < 8; t++) acc2[t] = mlp_f4{0.f, 0.f, 0.f, 0.f};
  const mlp_f4* wp4 = (const mlp_f4*)wp;
  mlp_f4 pre[4];
#pragma unroll
  for (int q = 0; q < 4; q++) sw[0][tid + 256 * q] = wp4[tid + 256 * q];
  __syncthreads();
  for (int c = 0; c < MLP_CHUNKS; c++) {
    const int cur = c & 1;
    if (c + 1 < MLP_CHUNKS) {
#pragma unroll
      for (int q = 0; q < 4; q++) pre[q] = wp4[(long)(c + 1) * (MLP_CHUNK_F / 4) + tid + 256 * q];
    }
    mlp_f4 acc1 = mlp_f4{0.f, 0.f, 0.f, 0.f}, acc1b = mlp_f4{0.f, 0.f, 0.f, 0.f};      // two accumulators: the dependent MFMA chain is half as long
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const mlp_f4 a4 = sw[cur][q * 64 + lane];
#pragma unroll
      for (int e = 0; e < 4; e++) if (4 * q + e < MLP_C / 4) {
        if (e & 1) acc1b = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[e], hk[4 * q + e], acc1b, 0, 0, 0);
        else acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[e], hk[4 * q + e], acc1, 0, 0, 0);
      }
    }
    acc1 += acc1b;
    float gv[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const float v = acc1[r] + b1[16 * c + 4 * g + r];
      gv[r] = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));      // nn.GELU() (exact)
    }
#pragma unroll
    for (int t = 0; t < 8; t++) {
      const mlp_f4 a4 = sw[cur][(8 + t) * 64 + lane];
#pragma unroll
      for (int e = 0; e < 4; e++) acc2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[e], gv[e], acc2[t], 0, 0, 0);
    }
    if (c + 1 < MLP_CHUNKS) {
#pragma unroll
      for (int q = 0; q < 4; q++) sw[cur ^ 1][tid + 256 * q] = pre[q];
    }
    __syncthreads();
  }
  if (!live) return;
#pragma unroll
  for (int t = 0; t < 8; t++) {
    const int col = 16 * t + 4 * g;
    if (col >= MLP_C) continue;
    const mlp_f4 xr = *(const mlp_f4*)(x + row * MLP_C + col), bb = *(const mlp_f4*)(b2 + col);
    *(mlp_f4*)(out + row * MLP_C + col) = xr + bb + acc2[t];
  }
}
// out[M][N] = (LayerNorm)(xin)[M][120] W^T + bias (+ resid): the linear layers of the DiffusionGPT block with 120 input features (query | key | value in
// one product, the attention output projection with the residual) on the f32 matrix cores.  Same operand scheme as the first product of the MLP kernel:
// D[n][row] per tile of 16 outputs (30 MFMA steps), the lane's four D registers are four consecutive outputs of its row - one float4 store.
__global__ __launch_bounds__(256) void k_linear120_f32(const float* __restrict__ xin, const float* __restrict__ wp, const float* __restrict__ bias, const float* __restrict__ resid,
                                                        float* __restrict__ out, long M, int N, const float* __restrict__ ln_w, const float* __restrict__ ln_b, float eps) {
  __shared__ mlp_f4 sw[2][8 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, g = lane >> 4;
  const long row = (long)blockIdx.x * 64 + wave * 16 + j;
  const bool live = row < M;
  const long rr = live ? row : (M - 1);
  float hk[MLP_C / 4];
  mlp_load_row(xin, rr, g, ln_w, ln_b, eps, hk);
  const int ntiles = (N + 15) / 16;
  const mlp_f4* wp4 = (const mlp_f4*)wp;
  mlp_f4 pre[2];
#pragma unroll
  for (int q = 0; q < 2; q++) sw[0][tid + 256 * q] = wp4[tid + 256 * q];
  __syncthreads();
  for (int t = 0; t < ntiles; t++) {
    const int cur = t & 1;
    if (t + 1 < ntiles) {
#pragma unroll
      for (int q = 0; q < 2; q++) pre[q] = wp4[(long)(t + 1) * 512 + tid + 256 * q];
    }
    mlp_f4 acc = mlp_f4{0.f, 0.f, 0.f, 0.f}, acc_b = mlp_f4{0.f, 0.f, 0.f, 0.f};      // two accumulators: the dependent MFMA chain is half as long
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const mlp_f4 a4 = sw[cur][q * 64 + lane];
#pragma unroll
      for (int e = 0; e < 4; e++) if (4 * q + e < MLP_C / 4) {
        if (e & 1) acc_b = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[e], hk[4 * q + e], acc_b, 0, 0, 0);
        else acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[e], hk[4 * q + e], acc, 0, 0, 0);
      }
    }
    const int col = 16 * t + 4 * g;
    if (live && col < N) {
      mlp_f4 v = acc + acc_b + *(const mlp_f4*)(bias + col);
      if (resid) v += *(const mlp_f4*)(resid + row * (long)N + col);
      *(mlp_f4*)(out + row * (long)N + col) = v;
    }
    if (t + 1 < ntiles) {
#pragma unroll
      for (int q = 0; q < 2; q++) sw[cur ^ 1][tid + 256 * q] = pre[q];
    }
    __syncthreads();
  }
}
// The whole sampling chain of the reference's DDPM policy (agents/models/diffusion/gc_diffusion.py:101-216 over diffusion_models.py:20-118 and
// common/mlp.py:9-46,114-182 as configs/agents/ddpm_agent.yaml + scripts/sorting_4/ddpm_benchmark.sh configure them: DiffusionMLP, hidden 256, 8 hidden layers =
// 4 pre-activation residual blocks, Mish, t_dim 8, window 1) in ONE kernel, f32 on the matrix cores.  Rows are independent: a workgroup owns 16 rows for all T
// denoising steps and all layers, its DD_NW waves each own DD_TPW of the 16 output tiles of every layer (wave w: outputs 16 DD_TPW w .. 16 DD_TPW (w + 1) - 1).
// D of v_mfma_f32_16x16x4_f32 (lane (g, j): outputs 4 g + r of row j) is a B operand of the next layer when step (t, r) sums the features {16 t + 4 g + r}, which only
// fixes how the next weight matrix is packed (d3il_amd/policies.py pack_ddpm_weights): [T_out][t][lane (g, i)][r] = W[16 T_out + i][16 t + 4 g + r].  So a layer is: every
// wave reads the 16 rows' 256 activations (16 float4 per lane) from LDS, runs DD_TPW x 64 MFMAs with its own share of the weights streamed from L2 (16 KB per tile,
// no sharing needed), applies bias / residual / Mish to its D registers and writes them to the other LDS buffer: one barrier per layer.
// The first layer (x | time embedding | state = 26 -> 28 inputs, 7 steps per tile) and the output layer (2 of 16 outputs; every wave computes it for the DDPM
// update of its copy of x) read their small packed matrices from L2.  The time embedding of step i is the same for every row (temb [T][8], evaluated once by the
// caller); the noise of all T + 1 draws comes as one tensor.
constexpr int DD_H = 256, DD_TILE_F4 = 16 * 64, DD_LAYER_F4 = 16 * DD_TILE_F4;      // float4 per packed output tile (16 KB), per packed layer
constexpr int DD_NW = 8, DD_TPW = 16 / DD_NW;      // waves per workgroup (16 rows), output tiles per wave: eight waves of two tiles - two waves per SIMD hide each other's weight loads
__device__ __forceinline__ float dd_mish(float x) {      // x tanh(softplus(x)), softplus with torch's threshold 20; tanh(log(1 + n)) = (n^2 + 2 n) / (n^2 + 2 n + 2), n = e^x
  if (x > 20.f) return x;
  const float n = expf(x), p = n * (n + 2.f);
  return x * (p / (p + 2.f));
}
// y[q] = (RES ? y[q] : 0) + bias + W m for the wave's four output tiles 4 w + q of one packed 256 x 256 layer (xin: the 16 rows' activations in LDS, B-operand order)
template <bool RES>
__device__ __forceinline__ void dd_layer4(const mlp_f4* __restrict__ wl, const float* __restrict__ bias, const mlp_f4* xin, mlp_f4* y, int w, int lane, int g) {
  mlp_f4 m[16], a[2][16];      // the weights of tile q + 1 are on their way from L2 while tile q is multiplied
  const mlp_f4* wt = wl + (long)(DD_TPW * w) * DD_TILE_F4 + lane;
#pragma unroll
  for (int t = 0; t < 16; t++) a[0][t] = wt[t * 64];
#pragma unroll
  for (int t = 0; t < 16; t++) m[t] = xin[t * 64 + lane];
#pragma unroll
  for (int q = 0; q < DD_TPW; q++) {
    const int To = DD_TPW * w + q;
    if (q + 1 < DD_TPW) {
#pragma unroll
      for (int t = 0; t < 16; t++) a[(q + 1) & 1][t] = wt[(q + 1) * DD_TILE_F4 + t * 64];
    }
    mlp_f4 acc[4];      // four accumulators: no MFMA waits for the one before it
    acc[0] = *(const mlp_f4*)(bias + 16 * To + 4 * g);
    acc[1] = RES ? y[q] : mlp_f4{0.f, 0.f, 0.f, 0.f};
    acc[2] = mlp_f4{0.f, 0.f, 0.f, 0.f}; acc[3] = mlp_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 16; t++)
#pragma unroll
      for (int r = 0; r < 4; r++) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q & 1][t][r], m[t][r], acc[r], 0, 0, 0);
    y[q] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  }
}
__global__ __launch_bounds__(64 * DD_NW) void k_ddpm_mlp_f32(const float* __restrict__ state, const float* __restrict__ noise, const float* __restrict__ temb, const float* __restrict__ w_in,
                                                       const float* __restrict__ b_in, const float* __restrict__ w_blk, const float* __restrict__ b_blk, const float* __restrict__ w_out,
                                                       const float* __restrict__ b_out, const float* __restrict__ sched, const float* __restrict__ bounds, float* __restrict__ out,
                                                       long n, int SD, int T, int nblk) {
  __shared__ mlp_f4 xb[2][DD_TILE_F4];      // the 16 rows' 256 activations, [t][lane] float4 = features 16 t + 4 g + r of row j: written by the layer that produces them, read by the next
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 15, g = lane >> 4;
  const long row = (long)blockIdx.x * 16 + j;
  const bool live = row < n;
  const long rr = live ? row : (n - 1);
  float st_k[7];      // the lane's share of the input row: feature 4 s + g of [x (2) | time embedding (8) | state (SD)]
#pragma unroll
  for (int s2 = 0; s2 < 7; s2++) { const int f = 4 * s2 + g; st_k[s2] = (f >= 10 && f < 10 + SD) ? state[rr * SD + f - 10] : 0.f; }
  const float lo0 = bounds[0], lo1 = bounds[1], hi0 = bounds[2], hi1 = bounds[3];
  float x0 = noise[rr * 2], x1 = noise[rr * 2 + 1];
  const mlp_f4* w_in4 = (const mlp_f4*)w_in;
  const mlp_f4* w_out4 = (const mlp_f4*)w_out;
  int buf = 0;
#pragma clang loop unroll(disable)
  for (int step = 0; step < T; step++) {
    const int i = T - 1 - step;
    float in_k[7];
#pragma unroll
    for (int s2 = 0; s2 < 7; s2++) { const int f = 4 * s2 + g; in_k[s2] = f == 0 ? x0 : (f == 1 ? x1 : (f < 10 ? temb[i * 8 + f - 2] : st_k[s2])); }
    mlp_f4 xo[DD_TPW];      // the wave's tiles of the residual stream
#pragma unroll
    for (int q = 0; q < DD_TPW; q++) {
      const int To = DD_TPW * w + q;
      mlp_f4 acc = *(const mlp_f4*)(b_in + 16 * To + 4 * g);
      const mlp_f4 a0 = w_in4[(To * 64 + lane) * 2], a1 = w_in4[(To * 64 + lane) * 2 + 1];
#pragma unroll
      for (int s2 = 0; s2 < 7; s2++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(s2 < 4 ? a0[s2] : a1[s2 - 4], in_k[s2], acc, 0, 0, 0);
      xo[q] = acc;
    }
#pragma clang loop unroll(disable)
    for (int b = 0; b < nblk; b++) {      // x + l2(mish(l1(mish(x))))
      mlp_f4 y[DD_TPW];
#pragma unroll
      for (int q = 0; q < DD_TPW; q++) xb[buf][(DD_TPW * w + q) * 64 + lane] = mlp_f4{dd_mish(xo[q][0]), dd_mish(xo[q][1]), dd_mish(xo[q][2]), dd_mish(xo[q][3])};
      __syncthreads();
      dd_layer4<false>((const mlp_f4*)w_blk + (long)(2 * b) * DD_LAYER_F4, b_blk + (2 * b) * DD_H, xb[buf], y, w, lane, g);
      buf ^= 1;
#pragma unroll
      for (int q = 0; q < DD_TPW; q++) xb[buf][(DD_TPW * w + q) * 64 + lane] = mlp_f4{dd_mish(y[q][0]), dd_mish(y[q][1]), dd_mish(y[q][2]), dd_mish(y[q][3])};
      __syncthreads();
      dd_layer4<true>((const mlp_f4*)w_blk + (long)(2 * b + 1) * DD_LAYER_F4, b_blk + (2 * b + 1) * DD_H, xb[buf], xo, w, lane, g);      // the residual rides in the accumulator
      buf ^= 1;
    }
#pragma unroll
    for (int q = 0; q < DD_TPW; q++) xb[buf][(DD_TPW * w + q) * 64 + lane] = xo[q];
    __syncthreads();
    mlp_f4 acc = mlp_f4{0.f, 0.f, 0.f, 0.f}, acc2 = mlp_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 16; t++) {
      const mlp_f4 a4 = w_out4[t * 64 + lane], m4 = xb[buf][t * 64 + lane];
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[0], m4[0], acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[1], m4[1], acc2, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[2], m4[2], acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[3], m4[3], acc2, 0, 0, 0);
    }
    buf ^= 1;
    acc += acc2;
    // epsilon of row j sits in lane (0, j), registers 0 and 1: every lane group of every wave takes it and does the same update on its copy of x
    const float e0 = __shfl(acc[0], j) + b_out[0], e1 = __shfl(acc[1], j) + b_out[1];
    const float sra = sched[i * 5], srm1 = sched[i * 5 + 1], c1 = sched[i * 5 + 2], c2 = sched[i * 5 + 3], sig = sched[i * 5 + 4];
    const float p0 = fminf(fmaxf(sra * x0 - srm1 * e0, lo0), hi0), p1 = fminf(fmaxf(sra * x1 - srm1 * e1, lo1), hi1);      // clipped x0 prediction
    const float z0 = noise[((long)(step + 1) * n + rr) * 2], z1 = noise[((long)(step + 1) * n + rr) * 2 + 1];
    x0 = (c1 * p0 + c2 * x0) + sig * z0;
    x1 = (c1 * p1 + c2 * x1) + sig * z1;
  }
  if (live && w == 0 && g == 0) { out[row * 2] = fminf(fmaxf(x0, lo0), hi0); out[row * 2 + 1] = fminf(fmaxf(x1, lo1), hi1); }
}
// The reference's ResidualMLPNetwork (agents/models/common/mlp.py:114-182: Linear, n pre-activation residual blocks x + l2(mish(l1(mish(x)))), Linear - the network of
// the BC agent, bc_agent.py:240-271, and the denoiser above without its sampling loop) in one launch, same operand scheme as k_ddpm_mlp_f32: 16 rows per
// workgroup, eight waves share the HID / 16 output tiles of a layer, the activations go from layer to layer through LDS in B-operand order.  HID = 128 or 256,
// at most 28 inputs and 16 outputs.
template <int HID, bool RES>
__device__ __forceinline__ void rm_layer(const mlp_f4* __restrict__ wl, const float* __restrict__ bias, const mlp_f4* xin, mlp_f4* y, int w, int lane, int g) {
  constexpr int NT = HID / 16, TPW = NT / 8;
  mlp_f4 m[NT];
#pragma unroll
  for (int t = 0; t < NT; t++) m[t] = xin[t * 64 + lane];
#pragma unroll
  for (int q = 0; q < TPW; q++) {
    const int To = TPW * w + q;
    const mlp_f4* wt = wl + (long)To * (NT * 64) + lane;
    mlp_f4 a[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) a[t] = wt[t * 64];
    mlp_f4 acc[4];
    acc[0] = *(const mlp_f4*)(bias + 16 * To + 4 * g);
    acc[1] = RES ? y[q] : mlp_f4{0.f, 0.f, 0.f, 0.f};
    acc[2] = mlp_f4{0.f, 0.f, 0.f, 0.f}; acc[3] = mlp_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
      for (int r = 0; r < 4; r++) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][r], m[t][r], acc[r], 0, 0, 0);
    y[q] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  }
}
template <int HID>
__global__ __launch_bounds__(512) void k_resmlp_f32(const float* __restrict__ x, const float* __restrict__ w_in, const float* __restrict__ b_in, const float* __restrict__ w_blk,
                                                     const float* __restrict__ b_blk, const float* __restrict__ w_out, const float* __restrict__ b_out, float* __restrict__ out,
                                                     long n, int IN, int OUT, int nblk) {
  constexpr int NT = HID / 16, TPW = NT / 8, LAYER_F4 = NT * NT * 64;
  __shared__ mlp_f4 xb[2][NT * 64];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 15, g = lane >> 4;
  const long row = (long)blockIdx.x * 16 + j;
  const bool live = row < n;
  const long rr = live ? row : (n - 1);
  float in_k[7];
#pragma unroll
  for (int s2 = 0; s2 < 7; s2++) { const int f = 4 * s2 + g; in_k[s2] = f < IN ? x[rr * IN + f] : 0.f; }
  const mlp_f4* w_in4 = (const mlp_f4*)w_in;
  const mlp_f4* w_out4 = (const mlp_f4*)w_out;
  mlp_f4 xo[TPW];
#pragma unroll
  for (int q = 0; q < TPW; q++) {
    const int To = TPW * w + q;
    mlp_f4 acc = *(const mlp_f4*)(b_in + 16 * To + 4 * g);
    const mlp_f4 a0 = w_in4[(To * 64 + lane) * 2], a1 = w_in4[(To * 64 + lane) * 2 + 1];
#pragma unroll
    for (int s2 = 0; s2 < 7; s2++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(s2 < 4 ? a0[s2] : a1[s2 - 4], in_k[s2], acc, 0, 0, 0);
    xo[q] = acc;
  }
  int buf = 0;
#pragma clang loop unroll(disable)
  for (int b = 0; b < nblk; b++) {
    mlp_f4 y[TPW];
#pragma unroll
    for (int q = 0; q < TPW; q++) xb[buf][(TPW * w + q) * 64 + lane] = mlp_f4{dd_mish(xo[q][0]), dd_mish(xo[q][1]), dd_mish(xo[q][2]), dd_mish(xo[q][3])};
    __syncthreads();
    rm_layer<HID, false>((const mlp_f4*)w_blk + (long)(2 * b) * LAYER_F4, b_blk + (2 * b) * HID, xb[buf], y, w, lane, g);
    buf ^= 1;
#pragma unroll
    for (int q = 0; q < TPW; q++) xb[buf][(TPW * w + q) * 64 + lane] = mlp_f4{dd_mish(y[q][0]), dd_mish(y[q][1]), dd_mish(y[q][2]), dd_mish(y[q][3])};
    __syncthreads();
    rm_layer<HID, true>((const mlp_f4*)w_blk + (long)(2 * b + 1) * LAYER_F4, b_blk + (2 * b + 1) * HID, xb[buf], xo, w, lane, g);
    buf ^= 1;
  }
#pragma unroll
  for (int q = 0; q < TPW; q++) xb[buf][(TPW * w + q) * 64 + lane] = xo[q];
  __syncthreads();
  if (w != 0) return;      // (after the last barrier) the output tile is one wave's work
  mlp_f4 acc = mlp_f4{0.f, 0.f, 0.f, 0.f}, acc2 = mlp_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < NT; t++) {
    const mlp_f4 a4 = w_out4[t * 64 + lane], m4 = xb[buf][t * 64 + lane];
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[0], m4[0], acc, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[1], m4[1], acc2, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[2], m4[2], acc, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[3], m4[3], acc2, 0, 0, 0);
  }
  acc += acc2;
  if (live) {
#pragma unroll
    for (int r = 0; r < 4; r++) if (4 * g + r < OUT) out[row * OUT + 4 * g + r] = acc[r] + b_out[4 * g + r];
  }
}
__global__ __launch_bounds__(256) void k_layernorm_f32(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ y,
                                                       long rows, int C, float eps) {
  const long row = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 5;
  const int sub = threadIdx.x & 31;
  const bool live = row < rows && 4 * sub < C;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live) v = reinterpret_cast<const float4*>(x + row * C)[sub];
  float s = v.x + v.y + v.z + v.w;
#pragma unroll
  for (int m = 16; m >= 1; m >>= 1) s += __shfl_xor(s, m);
  const float mean = s / (float)C;
  const float dx = live ? v.x - mean : 0.f, dy = live ? v.y - mean : 0.f, dz = live ? v.z - mean : 0.f, dw = live ? v.w - mean : 0.f;
  float q = dx * dx + dy * dy + dz * dz + dw * dw;
#pragma unroll
  for (int m = 16; m >= 1; m >>= 1) q += __shfl_xor(q, m);
  const float inv = rsqrtf(q / (float)C + eps);
  if (live) {
    const float4 ww = reinterpret_cast<const float4*>(w)[sub], bb = reinterpret_cast<const float4*>(b)[sub];
    reinterpret_cast<float4*>(y + row * C)[sub] = make_float4(dx * inv * ww.x + bb.x, dy * inv * ww.y + bb.y, dz * inv * ww.z + bb.z, dw * inv * ww.w + bb.w);
  }
}

template <int D4>      // D4 = D / 4 float4 chunks per head row (D a multiple of 4: 16-byte loads), or 0: scalar loads for any D <= 32
__global__ __launch_bounds__(256) void k_attention_causal_f32(const float* __restrict__ qkv, float* __restrict__ out, int B, int T, int H, int D) {
  // One lane serves TWO queries of a (sequence, head): i and T - 1 - i.  Query i attends to i + 1 keys, so the pair costs T + 1 key steps whatever i is:
  // the lanes of a wave run the same number of steps (one query per lane leaves half of the lane-steps idle under the causal mask).
  const int TP = (T + 1) / 2;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;     // ((b * H) + h) * TP + p: the query pairs of one (b, h) sit in adjacent lanes
  const long total = (long)B * H * TP;
  if (idx >= total) return;
  const int p = (int)(idx % TP), h = (int)((idx / TP) % H);
  const long b = idx / ((long)TP * H);
  const int C = H * D;
  const float* base = qkv + (b * T) * (long)(3 * C) + h * D;
  constexpr int DR = D4 > 0 ? 4 * D4 : 32;
  const float scale = 1.0f / sqrtf((float)D);
  auto load = [&](const float* ptr, float* r) {
    if constexpr (D4 > 0) {
#pragma unroll
      for (int c = 0; c < D4; c++) { const float4 t = reinterpret_cast<const float4*>(ptr)[c]; r[4 * c] = t.x; r[4 * c + 1] = t.y; r[4 * c + 2] = t.z; r[4 * c + 3] = t.w; }
    } else {
#pragma unroll
      for (int d = 0; d < 32; d++) r[d] = d < D ? ptr[d] : 0.0f;
    }
  };
  for (int half = 0; half < 2; half++) {
    const int i = half == 0 ? p : T - 1 - p;
    if (half == 1 && i == p) break;      // the middle query of an odd T
    float q[DR], acc[DR], kk[DR], vv[DR];
    load(base + (long)i * 3 * C, q);
#pragma unroll
    for (int d = 0; d < DR; d++) q[d] *= scale;
    // online softmax; key 0 is peeled (m = score_0, l = 1, acc = v_0) so that no infinity is ever formed: the device pass is built
    // with -ffinite-math-only, under which arithmetic on infinities is undefined
    load(base + C, kk); load(base + 2 * C, acc);
    float m = 0.0f, l = 1.0f;
#pragma unroll
    for (int d = 0; d < DR; d++) m += q[d] * kk[d];
    for (int j = 1; j <= i; j++) {
      const float* kj = base + (long)j * 3 * C + C;
      load(kj, kk); load(kj + C, vv);
      float sc = 0.0f;
#pragma unroll
      for (int d = 0; d < DR; d++) sc += q[d] * kk[d];
      const float mn = fmaxf(m, sc), corr = __expf(m - mn), pj = __expf(sc - mn);
      l = l * corr + pj;
#pragma unroll
      for (int d = 0; d < DR; d++) acc[d] = acc[d] * corr + pj * vv[d];
      m = mn;
    }
    float* o = out + (b * T + i) * (long)C + h * D;
    const float inv = 1.0f / l;
    if constexpr (D4 > 0) {
#pragma unroll
      for (int c = 0; c < D4; c++) reinterpret_cast<float4*>(o)[c] = make_float4(acc[4 * c] * inv, acc[4 * c + 1] * inv, acc[4 * c + 2] * inv, acc[4 * c + 3] * inv);
    } else {
#pragma unroll
      for (int d = 0; d < 32; d++) if (d < D) o[d] = acc[d] * inv;
    }
  }
}

// ---- random-policy harness (avoiding_sim.py:51-66 with a uniform random agent)
__device__ __forceinline__ void philox4x32_10(unsigned k0, unsigned k1, unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned* out) {
  const unsigned M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; r++) {
    unsigned hi0 = __umulhi(M0, c0), lo0 = M0 * c0, hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
    unsigned n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3; k0 += W0; k1 += W1;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__global__ void k_policy_begin(const unsigned char* __restrict__ mask, const double* __restrict__ state, double* __restrict__ des, int n, int stride) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n || (mask && !mask[e])) return;
#pragma unroll
  for (int k = 0; k < 3; k++) des[k * (size_t)stride + e] = state[(D3IL_STATE_TCP + k) * (size_t)stride + e];
}
__global__ void k_policy_action(double* __restrict__ des, double* __restrict__ actions, unsigned long long seed, unsigned long long env_offset,
                                unsigned t, int n, int stride) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  unsigned long long ge = env_offset + (unsigned long long)e;
  unsigned r[4];
  philox4x32_10((unsigned)seed, (unsigned)(seed >> 32), (unsigned)ge, (unsigned)(ge >> 32), t, 0u, r);
  // two uniforms in [0,1) with 32 random bits each -> delta in [-0.01, 0.01)
  double u0 = r[0] * (1.0 / 4294967296.0), u1 = r[1] * (1.0 / 4294967296.0);
  double x = des[e] + (0.02 * u0 - 0.01), y = des[(size_t)stride + e] + (0.02 * u1 - 0.01), z = des[2 * (size_t)stride + e];
  des[e] = x; des[(size_t)stride + e] = y;
  double* a = actions + (size_t)e * 7;
  a[0] = x; a[1] = y; a[2] = z; a[3] = 0; a[4] = 1; a[5] = 0; a[6] = 0;
}
// the same with the step counter t read from device memory (the captured form of the rollout step, d3il_random_rollout_step with option graph_rollout:
// a graph's kernel arguments are fixed at capture time); k_inc_counter, the last node of the graph, advances it
__global__ void k_policy_action_dev(double* __restrict__ des, double* __restrict__ actions, unsigned long long seed, unsigned long long env_offset,
                                    const unsigned* __restrict__ t_dev, int n, int stride) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const unsigned t = *t_dev;
  unsigned long long ge = env_offset + (unsigned long long)e;
  unsigned r[4];
  philox4x32_10((unsigned)seed, (unsigned)(seed >> 32), (unsigned)ge, (unsigned)(ge >> 32), t, 0u, r);
  double u0 = r[0] * (1.0 / 4294967296.0), u1 = r[1] * (1.0 / 4294967296.0);
  double x = des[e] + (0.02 * u0 - 0.01), y = des[(size_t)stride + e] + (0.02 * u1 - 0.01), z = des[2 * (size_t)stride + e];
  des[e] = x; des[(size_t)stride + e] = y;
  double* a = actions + (size_t)e * 7;
  a[0] = x; a[1] = y; a[2] = z; a[3] = 0; a[4] = 1; a[5] = 0; a[6] = 0;
}
__global__ void k_inc_counter(unsigned* __restrict__ t_dev) { if (threadIdx.x == 0 && blockIdx.x == 0) *t_dev += 1u; }
// Everything the random-policy harness does between two step launches, in ONE launch (d3il_random_rollout_step): the mask of the environments that finished
// (buf.last_reset), the episode counters and the per-context tally of the finished ones, their reset + re-latch (k_avoiding_auto_reset), and the policy's NEXT
// action for every environment (k_policy_action with step counter t_next; a reset lane draws from its re-latched pose, as in the separate sequence).  A
// rollout step is then two launches - step kernel, this kernel - instead of five and a copy.
__global__ __launch_bounds__(WAVE) void k_avoiding_tail(const double* __restrict__ init_qpos, double* __restrict__ state, unsigned* __restrict__ flags, int* __restrict__ steps,
                                                        float* __restrict__ obs, unsigned char* __restrict__ done, unsigned char* __restrict__ success,
                                                        unsigned short* __restrict__ mode, double* __restrict__ des, long long* __restrict__ episode_counts,
                                                        unsigned char* __restrict__ mask, const int* __restrict__ ctx_id, long long* __restrict__ table, int n_ctx,
                                                        double* __restrict__ actions, double* __restrict__ des_before, unsigned long long seed, unsigned long long env_offset, unsigned t_next, int n, int stride) {
  int e = blockIdx.x * WAVE + threadIdx.x;
  if (e >= n) return;
  const bool fin = done[e] != 0;
  mask[e] = fin ? 1 : 0;
  if (fin) {
    const bool ok = success[e] != 0;
    atomicAdd((unsigned long long*)&episode_counts[0], 1ull);
    if (ok) atomicAdd((unsigned long long*)&episode_counts[1], 1ull);
    if (table) {
      int c = ctx_id ? ctx_id[e] : 0;
      if (c >= 0 && c < n_ctx) {
        long long* row = table + (size_t)c * D3IL_TALLY_ROW;
        atomicAdd((unsigned long long*)&row[0], 1ull);
        const int code = (int)(short)mode[e];
        if (ok) {
          atomicAdd((unsigned long long*)&row[1], 1ull);
          if (code >= 0 && code < D3IL_TALLY_ROW - 2) atomicAdd((unsigned long long*)&row[2 + code], 1ull);
        }
      }
    }
    EnvState st;
    double iq[NARM];
#pragma unroll
    for (int k = 0; k < NARM; k++) iq[k] = init_qpos[k];
    float o[2];
    env_reset(kAvoidingConsts, st, iq, o);
    store_state(state, flags, steps, stride, e, st);
    store_outputs(st, e, o, 0, obs, done, success, mode);
#pragma unroll
    for (int k = 0; k < 3; k++) des[k * (size_t)stride + e] = st.tcp[k];
  }
  unsigned long long ge = env_offset + (unsigned long long)e;
  unsigned r[4];
  philox4x32_10((unsigned)seed, (unsigned)(seed >> 32), (unsigned)ge, (unsigned)(ge >> 32), t_next, 0u, r);
  double u0 = r[0] * (1.0 / 4294967296.0), u1 = r[1] * (1.0 / 4294967296.0);
  des_before[e] = des[e]; des_before[(size_t)stride + e] = des[(size_t)stride + e];      // the pose this draw starts from: put back if the sequence is interrupted
  double x = des[e] + (0.02 * u0 - 0.01), y = des[(size_t)stride + e] + (0.02 * u1 - 0.01), z = des[2 * (size_t)stride + e];
  des[e] = x; des[(size_t)stride + e] = y;
  double* a = actions + (size_t)e * 7;
  a[0] = x; a[1] = y; a[2] = z; a[3] = 0; a[4] = 1; a[5] = 0; a[6] = 0;
}
__global__ void k_restore_des(double* __restrict__ des, const double* __restrict__ des_before, int n, int stride) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  des[e] = des_before[e]; des[(size_t)stride + e] = des_before[(size_t)stride + e];
}

__global__ void k_count_metrics(const unsigned char* __restrict__ done, const unsigned* __restrict__ flags, long long* __restrict__ counts, int n) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  if (done[e]) atomicAdd((unsigned long long*)&counts[0], 1ull);
  if (flags[e] & F_SUCCESS) {
    atomicAdd((unsigned long long*)&counts[1], 1ull);
    atomicAdd((unsigned long long*)&counts[2 + (flags[e] & F_MODE_MASK)], 1ull);
  }
}

// Episode tally of the rollout harnesses (avoiding_sim.py:45-54, pushing_sim.py:43-86, sorting_sim.py:100-133 loop over contexts and
// trajectories and record success / mode of the step that returned done): every finished environment adds to row ctx_id[e] of an
// int64 table [n_ctx][D3IL_TALLY_ROW]: [0] episodes, [1] successes, [2 + code] successes by mode code (Avoiding: 9-bit code,
// Pushing: info['mode'] + 1, Sorting: np.packbits code, Stacking: n | c0 << 2 | c1 << 4 | c2 << 6), and for Stacking
// [2 + D3IL_TALLY_ALL + code] ALL finished episodes by code.  Integer sums: bit-exact and order independent.
__global__ void k_episode_tally(const unsigned char* __restrict__ done, const unsigned char* __restrict__ success, const unsigned short* __restrict__ mode,
                                const int* __restrict__ ctx_id, long long* __restrict__ table, long long* __restrict__ episode_counts, int n, int n_ctx, int mode_bias, int all_by_code) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n || !done[e]) return;
  if (episode_counts) {
    atomicAdd((unsigned long long*)&episode_counts[0], 1ull);
    if (success[e]) atomicAdd((unsigned long long*)&episode_counts[1], 1ull);
  }
  if (!table) return;
  int c = ctx_id ? ctx_id[e] : 0;
  if (c < 0 || c >= n_ctx) return;
  long long* row = table + (size_t)c * D3IL_TALLY_ROW;
  atomicAdd((unsigned long long*)&row[0], 1ull);
  int code = (int)(short)mode[e] + mode_bias;
  if (success[e]) {
    atomicAdd((unsigned long long*)&row[1], 1ull);
    if (code >= 0 && code < D3IL_TALLY_ROW - 2) atomicAdd((unsigned long long*)&row[2 + code], 1ull);
  }
  // Stacking: info['success_1'] / ['success_2'] (stacking_sim.py:118-136) belong to episodes that did NOT stack all three boxes too,
  // so every finished episode is also counted by its order code (number of letters + colours, < 256) in the upper half of the row
  if (all_by_code && code >= 0 && code < 256) atomicAdd((unsigned long long*)&row[2 + D3IL_TALLY_ALL + code], 1ull);
}
// contexts of the last reset of every environment (what auto-reset starts the next trajectory of that lane from)
__global__ void k_store_contexts(const unsigned char* __restrict__ mask, const double* __restrict__ src, double* __restrict__ dst, int n, int dim) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * dim) return;
  if (mask && !mask[i / dim]) return;
  dst[i] = src[i];
}

}  // namespace d3il

// ====================================================================== C ABI
using namespace d3il;

// the public header documents the layouts the engines define: keep them in step
static_assert(D3IL_STACK_STATE_BOX == SK_STATE_BOX && D3IL_STACK_STATE_WARM == SK_STATE_WARM && D3IL_STACK_STATE_F64 == SK_STATE_F64, "d3il_rollout.h: Stacking state layout");
static_assert(D3IL_SFLAG_WARM_VALID == SKF_WARM_VALID && D3IL_SFLAG_HAND_NEAR == SKF_HAND_NEAR && D3IL_PFLAG_CON_OVERFLOW == SKF_CON_OVERFLOW && D3IL_PFLAG_OFF_TABLE == SKF_OFF_TABLE,
              "d3il_rollout.h: Stacking flag bits");
static_assert(D3IL_SFLAG_MODE_MASK == (SKF_NMODE_MASK | (0x3Fu << SKF_IND_SHIFT)), "d3il_rollout.h: Stacking order code");
static_assert(D3IL_ALIGN_STATE_BOX == AL_STATE_BOX && D3IL_ALIGN_STATE_WARM == AL_STATE_WARM && D3IL_ALIGN_STATE_TARGET == AL_STATE_TARGET && D3IL_ALIGN_STATE_F64 == AL_STATE_F64, "d3il_rollout.h: Aligning state layout");
static_assert(D3IL_INS_STATE_BOX == 42 && D3IL_INS_STATE_WARM == 42 + 13 * 3 && D3IL_INS_STATE_TASK == 42 + 13 * 3 + 6 * 3 + NDOF && D3IL_INS_STATE_F64 == gen_state_rows(3), "d3il_rollout.h: Inserting state layout");
static_assert(D3IL_PUSH_STATE_BOX == 42 && D3IL_PUSH_STATE_WARM == 42 + 13 * 2 && D3IL_PUSH_STATE_TASK == 42 + 13 * 2 + 6 * 2 + NDOF && D3IL_PUSH_STATE_F64 == gen_state_rows(2) && D3IL_TALLY_ALL + 256 <= D3IL_TALLY_ROW - 2,
              "d3il_rollout.h: Pushing state rows / tally row");

static inline bool gen_task(int task_id) {      // the tasks of the generic engine (gen_step.h): Pushing = its two cubes, the table slabs AND the frame beams as static boxes
  return task_id == D3IL_TASK_SORTING || task_id == D3IL_TASK_INSERTING || task_id == D3IL_TASK_PUSHING;
}

constexpr int RG_SLOTS = 2, RG_SAMPLE = 8;
struct d3il_handle_s {
  int task_id, n, stride, device;
  PandaConsts hc;          // host copy
  PandaConsts* dc;         // device copy
  GenConsts gc;            // Sorting / Inserting / Pushing: cubes, static boxes, contact parameter sets (host copy; the device copy is the __constant__ object)
  StackConsts kc;          // Stacking: boxes, finger geoms, contact parameter sets (host copy; device: __constant__)
  AlignTask atk;           // Aligning: success / mode thresholds (device: __constant__ g_align_task)
  double* d_scratch;       // contact records (generic engine: GG_BLOCK doubles per environment; cooperative engine: SG_SIZE)
  int state_rows;          // f64 state fields per environment (42 Avoiding, 91 Pushing, ...: d3il_buffers.state_rows)
  double* d_init_qpos;
  bool started;
  d3il_buffers buf;
  bool fast, timing;
  int split;              // -1 auto, 0 fused single-wave kernel, 1 two-wave (controller || physics) kernel
  bool serve_avail;       // the device grants the LDS the three-wave form needs
  int serve_max_wg;       // the split kernel runs with its third wave (rare constraint paths) up to this many workgroups (one per CU); 0: never
  int lanes;              // active lanes (environments) per wave: 64, or fewer to spread a small batch over more SIMDs
  int lds_pad;            // dynamic LDS bytes requested per workgroup: spreads the single-wave workgroups over CUs
  hipEvent_t ev0, ev1;     // the event pair of the LAST timed launch (aliases of a ring slot)
  bool ev_valid, ev_created;
  // every timed launch gets its own event pair from a ring; a pair is read (and its time accumulated) when its slot comes round again or when
  // d3il_timing_stats drains the ring - so the host never waits for a launch it has just enqueued
  hipEvent_t ring0[128], ring1[128];
  bool ring_created;
  long ring_head, ring_drained, t_n;
  double t_sum, t_min, t_max;
  int tol_mode;            // 0 production stopping rule of the contact solvers, 1 the oracle's (solver_strict)
  int stack_reset_coop;    // Stacking: 1 (default) env.reset() runs through the step kernel's cooperative phases, 0 the one-lane reset kernel
  unsigned long long kc_id; // identity of this handle's StackConsts in the device's g_stack_consts cache (Stacking, Aligning)
  double* d_ctx;           // [n][ctx_dim] context of the last reset of every environment (Pushing 14, Sorting 7 nb)
  int ctx_dim;
  uint8_t* d_mask;         // [stride] environments reset by the last d3il_auto_reset (buf.last_reset)
  const int32_t* tally_ctx; int tally_nctx; int64_t* tally_table;   // caller-owned device memory (d3il_set_tally)
  // captured rollout step (option graph_rollout; Avoiding random-policy harness): RG_SLOTS instantiated graphs of [policy, step, auto-reset, counter + 1],
  // launched round robin - one runtime call per step instead of eight
  bool rg_enabled, rg_ready, rg_capturing, rg_ev_made, rg_timing;
  int rg_slot;
  hipGraph_t rg_graph[RG_SLOTS]; hipGraphExec_t rg_exec[RG_SLOTS];
  unsigned* rg_t_dev; uint32_t rg_next_t; long rg_launched, rg_drained;
  uint64_t rg_seed, rg_off; double* rg_actions; int64_t* rg_counts; hipStream_t rg_stream;
  // fused tail of the Avoiding rollout step (k_avoiding_tail): the next step's action is already in `actions` when these match the next call
  bool prep_valid; uint32_t prep_t; uint64_t prep_seed, prep_off; double* prep_actions; bool fuse_tail; double* d_des_before; hipStream_t prep_stream;
  bool info_is_view;       // buf.info_f64 points into buf.state (Pushing on the generic engine: its two task rows) - not freed on its own
};

// The Stacking kernels read their model from one __constant__ object per device (scalar loads, no pointer across call boundaries).  Stacking handles on one
// device therefore have to share the model: d3il_create refuses a different one while another handle is alive (a second model would silently
// re-point the kernels of the first).  Handles of the generic engine reload the constants per launch (GenLaunch).
struct ActiveModel { int refs; bool valid; GenConsts gc; StackConsts kc; };
static ActiveModel g_active_gen[16], g_active_stack[16];
static int g_active_tol[16];   // solver tolerance set currently in the device's g_solver_tol (0 production)
// g_active_* / g_active_tol are process-global (one __constant__ object per device): every read-modify-write of them - d3il_create,
// d3il_destroy, the solver-rule switch - holds this mutex (two host threads with one handle each raced on them)
static std::mutex g_model_mutex;

// g_stack_consts (one __constant__ object per device) serves the Stacking engine AND its Aligning variant: it is a cache of the constants of
// the handle that launched last.  A handle with other constants reloads it before its launch, fenced by device synchronisations (handles
// with different engine constants must not have kernels in flight on one device at the same time - they never share a stream in practice).
static unsigned long long g_stack_loaded_id[16];
static StackConsts g_stack_loaded[16];
static unsigned long long g_kc_counter = 0;

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(D3IL_EHIP, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

extern "C" {

const char* d3il_last_error(void) { return g_err.c_str(); }
size_t d3il_blob_sizeof(void) { return sizeof(d3il_model_blob); }
int d3il_version(void) { return 2; }

static void rg_drop(d3il_handle_s* h);
static void rg_drop_for_free(d3il_handle_s* h) {
  if (h->rg_ready) for (int k = 0; k < RG_SLOTS; k++) { (void)hipGraphExecDestroy(h->rg_exec[k]); (void)hipGraphDestroy(h->rg_graph[k]); }
  if (h->rg_t_dev) (void)hipFree(h->rg_t_dev);
}
static void free_handle(d3il_handle_s* h) {
  if (!h) return;
  int dev = h->device;
  if (dev >= 0 && dev < 16) {
    std::lock_guard<std::mutex> lock(g_model_mutex);
    if (gen_task(h->task_id) && g_active_gen[dev].refs > 0) g_active_gen[dev].refs--;
    if (h->task_id == D3IL_TASK_STACKING && g_active_stack[dev].refs > 0) g_active_stack[dev].refs--;
  }
  void* ptrs[] = {h->dc, h->d_init_qpos, h->buf.obs, h->buf.done, h->buf.success, h->buf.mode, h->buf.state, h->buf.flags, h->buf.step_count, h->buf.policy_des,
                  h->info_is_view ? nullptr : (void*)h->buf.info_f64, h->d_scratch, h->d_ctx, h->d_mask, h->d_des_before};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  if (h->ring_created) for (int i = 0; i < 128; i++) { (void)hipEventDestroy(h->ring0[i]); (void)hipEventDestroy(h->ring1[i]); }
  rg_drop_for_free(h);
  delete h;
}
// inside d3il_create: every failure releases what has been allocated so far
#define HIPCHK_H(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { free_handle(h); return fail(D3IL_EHIP, std::string(#x) + ": " + hipGetErrorString(e_)); } } while (0)

int d3il_create(int task_id, int n_envs, int device_id, const void* model_blob, size_t blob_len, d3il_handle* out) {
  if (!out || !model_blob) return fail(D3IL_EINVAL, "d3il_create: null argument");
  *out = nullptr;
  if (blob_len != sizeof(d3il_model_blob)) return fail(D3IL_EBLOB, "d3il_create: blob size mismatch");
  if (n_envs <= 0) return fail(D3IL_EINVAL, "d3il_create: n_envs must be positive");
  const d3il_model_blob& m = *(const d3il_model_blob*)model_blob;
  if (task_id != m.task_id) return fail(D3IL_EINVAL, "d3il_create: task_id does not match the model blob");
  if (task_id != D3IL_TASK_AVOIDING && task_id != D3IL_TASK_PUSHING && task_id != D3IL_TASK_SORTING && task_id != D3IL_TASK_STACKING && task_id != D3IL_TASK_ALIGNING && task_id != D3IL_TASK_INSERTING)
    return fail(D3IL_EUNSUPPORTED, "d3il_create: unknown task id (Avoiding, Pushing, Sorting, Stacking, Aligning and Inserting are implemented)");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(D3IL_ENODEVICE, "d3il_create: no HIP device available (there is no CPU fallback)");
  if (device_id < 0 || device_id >= ndev || device_id >= 16) return fail(D3IL_ENODEVICE, "d3il_create: device_id out of range");
  HIPCHK(hipSetDevice(device_id));
  d3il_handle_s* h = new d3il_handle_s();
  std::memset(&h->buf, 0, sizeof h->buf);
  h->task_id = -1; h->device = device_id;      // task_id is set once the model reference is taken (free_handle)
  h->dc = nullptr; h->d_init_qpos = nullptr; h->d_scratch = nullptr; h->d_ctx = nullptr; h->d_mask = nullptr; h->ev_created = false; h->ring_created = false;
  h->tally_ctx = nullptr; h->tally_nctx = 0; h->tally_table = nullptr; h->tol_mode = 0; h->ctx_dim = 0; h->stack_reset_coop = 1; h->kc_id = 0;
  const char* err = "";
  int rc = build_panda_consts(m, h->hc, &err);
  if (rc) { free_handle(h); return fail(D3IL_EBLOB, std::string("d3il_create: ") + err); }
  finish_invweights(h->hc);
  if (gen_task(task_id) && build_gen_consts(m, h->hc, h->gc, &err)) { free_handle(h); return fail(D3IL_EBLOB, std::string("d3il_create: ") + err); }
  if (task_id == D3IL_TASK_STACKING && build_stack_consts(m, h->hc, h->kc, &err)) { free_handle(h); return fail(D3IL_EBLOB, std::string("d3il_create: ") + err); }
  if (task_id == D3IL_TASK_ALIGNING && build_coop_align_consts(m, h->hc, h->kc, h->atk, &err)) { free_handle(h); return fail(D3IL_EBLOB, std::string("d3il_create: ") + err); }
  {  // the kernels are specialised at build time to the robot model (csrc/gen/avoiding_consts.inc): the runtime blob
     // must describe the same arm, controller and (Avoiding) obstacles.  n_substeps / max_steps stay run-time parameters.
    PandaConsts a = h->hc, b = task_id == D3IL_TASK_STACKING ? kStackingConsts : kAvoidingConsts;   // Stacking: the gripper robot without the rod
    if (task_id != D3IL_TASK_AVOIDING) {   // no obstacles, other task constants: only the arm / controller part is compared
      a.n_obst = b.n_obst;
      std::memcpy(a.ob_c, b.ob_c, sizeof a.ob_c); std::memcpy(a.ob_u, b.ob_u, sizeof a.ob_u); std::memcpy(a.ob_r, b.ob_r, sizeof a.ob_r); std::memcpy(a.ob_h, b.ob_h, sizeof a.ob_h);
      std::memcpy(a.ct_K, b.ct_K, sizeof a.ct_K); std::memcpy(a.ct_B, b.ct_B, sizeof a.ct_B); std::memcpy(a.ct_solimp, b.ct_solimp, sizeof a.ct_solimp);
      std::memcpy(a.ct_margin, b.ct_margin, sizeof a.ct_margin); std::memcpy(a.ct_fric, b.ct_fric, sizeof a.ct_fric); std::memcpy(a.task_f, b.task_f, sizeof a.task_f);
    }
    bool same = a.n_obst == b.n_obst && a.ik_iters == b.ik_iters;
    a.n_obst = b.n_obst = 0; a.ik_iters = b.ik_iters = 0; a.n_substeps = b.n_substeps = 0; a.max_steps = b.max_steps = 0; a.pad_i = b.pad_i = 0; a.pad_j = b.pad_j = 0;
    const double* pa = (const double*)&a; const double* pb = (const double*)&b;
    for (size_t i = 0; same && i < sizeof(PandaConsts) / sizeof(double); i++) {
      double d = pa[i] - pb[i], m = pa[i] < 0 ? -pa[i] : pa[i];
      if (!(d <= 1e-12 * (m > 1 ? m : 1) && -d <= 1e-12 * (m > 1 ? m : 1))) same = false;
    }
    if (!same) { free_handle(h); return fail(D3IL_EUNSUPPORTED, "d3il_create: the model blob differs from the model this library was specialised for at build time; "
                                                                 "regenerate csrc/gen/*_consts.inc and rebuild (python -m d3il_amd.build)"); }
  }
  const bool sorting = gen_task(task_id);       // Sorting, Inserting and Pushing run on the generic engine (gen_step.h)
  const bool gen_pushing = sorting && task_id == D3IL_TASK_PUSHING;
  // (handles of the generic engine with different models - Sorting-2 / 4, Inserting, Pushing - may live side by side: every launch checks the constants the
  // device holds and reloads them when they are another handle's, sync_gen_consts_locked below)
  const bool stacking = task_id == D3IL_TASK_STACKING;
  if (stacking) {
    ActiveModel& am = g_active_stack[device_id];
    bool other;
    { std::lock_guard<std::mutex> lock(g_model_mutex); other = am.refs > 0 && std::memcmp(&am.kc, &h->kc, sizeof(StackConsts)) != 0; }
    if (other) {
      free_handle(h);
      return fail(D3IL_EUNSUPPORTED, "d3il_create: another live Stacking handle on this device uses a different model (the kernels read one model per device from constant memory); destroy it first");
    }
  }
  h->n = n_envs; h->stride = (n_envs + WAVE - 1) / WAVE * WAVE;
  h->started = false; h->split = -1; h->serve_avail = true; h->serve_max_wg = 256; h->lanes = WAVE; h->lds_pad = -1; h->fast = true; h->timing = false; h->ev_valid = false;
  const bool aligning = task_id == D3IL_TASK_ALIGNING;
  h->state_rows = (sorting ? gen_state_rows(h->gc.nb) : (stacking ? SK_STATE_F64 : (aligning ? AL_STATE_F64 : D3IL_STATE_F64)));
  h->ctx_dim = (sorting ? 7 * h->gc.nb : (stacking ? 21 : (aligning ? AL_CTX : 0)));
  size_t S = (size_t)h->stride;
  d3il_buffers& b = h->buf;
  b.n_envs = n_envs; b.stride = h->stride; b.obs_dim = sorting ? 2 + 3 * h->gc.nb : (stacking ? SK_OBS : (aligning ? AL_OBS : 2)); b.action_dim = stacking ? SK_ACT : 7; b.state_rows = h->state_rows; b.n_info_f64 = (aligning || gen_pushing) ? 2 : (stacking ? 1 : 0);
  HIPCHK_H(hipMalloc(&h->dc, sizeof(PandaConsts)));
  if (task_id == D3IL_TASK_AVOIDING) {
    // the three-wave form needs AVOID_LDS_SERVE of dynamic LDS on top of its static LDS; a device that cannot grant it runs the two-wave form
    if (hipFuncSetAttribute((const void*)k_avoiding_step_split<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)AVOID_LDS_SERVE) != hipSuccess) {
      (void)hipGetLastError();
      h->serve_avail = false; h->serve_max_wg = 0;
    }
  }
  HIPCHK_H(hipMemcpy(h->dc, &h->hc, sizeof(PandaConsts), hipMemcpyHostToDevice));
  HIPCHK_H(hipMalloc(&h->d_init_qpos, 7 * sizeof(double)));
  HIPCHK_H(hipMalloc(&b.obs, S * b.obs_dim * sizeof(float)));
  HIPCHK_H(hipMalloc(&b.done, S)); HIPCHK_H(hipMalloc(&b.success, S));
  HIPCHK_H(hipMalloc(&b.mode, S * sizeof(uint16_t)));
  HIPCHK_H(hipMalloc(&b.state, S * h->state_rows * sizeof(double)));
  HIPCHK_H(hipMalloc(&b.flags, S * sizeof(uint32_t)));
  HIPCHK_H(hipMalloc(&b.step_count, S * sizeof(int32_t)));
  HIPCHK_H(hipMalloc(&b.policy_des, S * 3 * sizeof(double)));
  HIPCHK_H(hipMalloc(&h->d_mask, S)); HIPCHK_H(hipMemset(h->d_mask, 0, S));
  b.last_reset = h->d_mask;
  HIPCHK_H(hipMemset(b.obs, 0, S * b.obs_dim * sizeof(float))); HIPCHK_H(hipMemset(b.done, 0, S)); HIPCHK_H(hipMemset(b.success, 0, S));
  HIPCHK_H(hipMemset(b.mode, 0, S * sizeof(uint16_t))); HIPCHK_H(hipMemset(b.state, 0, S * h->state_rows * sizeof(double)));
  HIPCHK_H(hipMemset(b.flags, 0, S * sizeof(uint32_t))); HIPCHK_H(hipMemset(b.step_count, 0, S * sizeof(int32_t)));
  HIPCHK_H(hipMemset(b.policy_des, 0, S * 3 * sizeof(double)));
  if (h->ctx_dim) { HIPCHK_H(hipMalloc(&h->d_ctx, S * h->ctx_dim * sizeof(double))); HIPCHK_H(hipMemset(h->d_ctx, 0, S * h->ctx_dim * sizeof(double))); }
  if (sorting) {
    {
      std::lock_guard<std::mutex> lock(g_model_mutex);
      g_active_gen[device_id].refs++; h->task_id = task_id;      // the constants go to the device with the first launch (GenLaunch)
    }
    HIPCHK_H(hipMalloc(&h->d_scratch, S * GG_BLOCK * sizeof(double))); HIPCHK_H(hipMemset(h->d_scratch, 0, S * GG_BLOCK * sizeof(double)));
    if (gen_pushing) { b.info_f64 = b.state + (size_t)(gen_state_rows(h->gc.nb) - 2) * S; h->info_is_view = true; }      // info['mean_distance'], reward: the task rows of the state buffer (gen_step.h gpush_*)
    HIPCHK_H(hipFuncSetAttribute((const void*)k_sorting_step<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, GEN_LDS_STEP));
    HIPCHK_H(hipFuncSetAttribute((const void*)k_sorting_step<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, GEN_LDS_STEP));
    HIPCHK_H(hipFuncSetAttribute((const void*)k_sorting_step<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, GEN_LDS_STEP));
    HIPCHK_H(hipFuncSetAttribute((const void*)k_sorting_step<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, GEN_LDS_STEP));
    HIPCHK_H(hipFuncSetAttribute((const void*)k_sorting_reset, hipFuncAttributeMaxDynamicSharedMemorySize, GEN_LDS_H));
  }
  if (stacking) {
    ActiveModel& am = g_active_stack[device_id];
    {
      std::unique_lock<std::mutex> lock(g_model_mutex);
      if (am.refs == 0) am.kc = h->kc;
      am.refs++; h->task_id = task_id;
      h->kc_id = ++g_kc_counter;      // loaded into g_stack_consts by the first launch (sync_stack_consts)
    }
    HIPCHK_H(hipMalloc(&b.info_f64, S * sizeof(double))); HIPCHK_H(hipMemset(b.info_f64, 0, S * sizeof(double)));
    HIPCHK_H(hipMalloc(&h->d_scratch, S * SG_SIZE * sizeof(double))); HIPCHK_H(hipMemset(h->d_scratch, 0, S * SG_SIZE * sizeof(double)));
    HIPCHK_H(hipFuncSetAttribute((const void*)k_stacking_step, hipFuncAttributeMaxDynamicSharedMemorySize, STACK_LDS));
    HIPCHK_H(hipFuncSetAttribute((const void*)k_stacking_reset, hipFuncAttributeMaxDynamicSharedMemorySize, STACK_LDS_RESET));
  }
  if (aligning) {
    {   // the engine constants go through the g_stack_consts cache like the Stacking ones; the task thresholds have their own object
      std::lock_guard<std::mutex> lock(g_model_mutex);
      h->kc_id = ++g_kc_counter;
    }
    HIPCHK_H(hipMalloc(&b.info_f64, S * 2 * sizeof(double))); HIPCHK_H(hipMemset(b.info_f64, 0, S * 2 * sizeof(double)));
    HIPCHK_H(hipMalloc(&h->d_scratch, S * SG_SIZE * sizeof(double))); HIPCHK_H(hipMemset(h->d_scratch, 0, S * SG_SIZE * sizeof(double)));
    HIPCHK_H(hipFuncSetAttribute((const void*)k_aligning_step, hipFuncAttributeMaxDynamicSharedMemorySize, STACK_LDS));
  }
  h->task_id = task_id;
  h->ring_created = false; h->ring_head = h->ring_drained = h->t_n = 0; h->t_sum = 0; h->t_min = 1e300; h->t_max = 0;
  h->ev_created = true;
  *out = h;
  return D3IL_OK;
}

int d3il_destroy(d3il_handle h) {
  if (!h) return fail(D3IL_EINVAL, "d3il_destroy: null handle");
  (void)hipSetDevice(h->device);
  (void)hipDeviceSynchronize();
  free_handle(h);
  return D3IL_OK;
}

int d3il_start(d3il_handle h, const double* init_qpos7) {
  if (!h || !init_qpos7) return fail(D3IL_EINVAL, "d3il_start: null argument");
  for (int k = 0; k < 7; k++) if (!(init_qpos7[k] == init_qpos7[k])) return fail(D3IL_EINVAL, "d3il_start: init_qpos contains NaN");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipMemcpy(h->d_init_qpos, init_qpos7, 7 * sizeof(double), hipMemcpyHostToDevice));
  h->started = true;
  return D3IL_OK;
}

// Loads this handle's engine constants into the device's g_stack_consts if another handle's are there.  The caller HOLDS g_model_mutex (StackLaunch
// below) from here until its kernel launch has been enqueued: otherwise a second host thread whose handle has other constants could reload the object
// between this handle's check and its launch.  A reload waits for everything in flight on the device first (hipDeviceSynchronize), so a
// kernel already launched keeps the constants it was launched with.
static AlignTask g_align_loaded[16];
static bool g_align_loaded_valid[16];
static int sync_stack_consts_locked(d3il_handle_s* h) {
  if (h->task_id == D3IL_TASK_ALIGNING && (!g_align_loaded_valid[h->device] || std::memcmp(&g_align_loaded[h->device], &h->atk, sizeof(AlignTask)) != 0)) {
    // the Aligning thresholds are a per-device __constant__ object like the engine constants: a handle whose blob carries other thresholds reloads them
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_align_task), &h->atk, sizeof(AlignTask)));
    HIPCHK(hipDeviceSynchronize());
    g_align_loaded[h->device] = h->atk; g_align_loaded_valid[h->device] = true;
  }
  if (g_stack_loaded_id[h->device] == h->kc_id) return D3IL_OK;
  if (g_stack_loaded_id[h->device] != 0 && std::memcmp(&g_stack_loaded[h->device], &h->kc, sizeof(StackConsts)) == 0) { g_stack_loaded_id[h->device] = h->kc_id; return D3IL_OK; }
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_stack_consts), &h->kc, sizeof(StackConsts)));
  HIPCHK(hipDeviceSynchronize());
  g_stack_loaded[h->device] = h->kc; g_stack_loaded_id[h->device] = h->kc_id;
  return D3IL_OK;
}
struct StackLaunch {      // scope = "constants checked ... kernel enqueued"
  std::unique_lock<std::mutex> lock;
  int rc;
  explicit StackLaunch(d3il_handle_s* h) : lock(g_model_mutex), rc(sync_stack_consts_locked(h)) {}
};

// g_gen_consts (one __constant__ object per device) is a cache of the generic engine's constants of the handle that launched last, like g_stack_consts:
// a handle with another model (Sorting-2 / 4, Inserting, Pushing) reloads it before its launch, fenced by device synchronisations.
static GenConsts g_gen_loaded[16];
static bool g_gen_loaded_valid[16];
static int sync_gen_consts_locked(d3il_handle_s* h) {
  if (g_gen_loaded_valid[h->device] && std::memcmp(&g_gen_loaded[h->device], &h->gc, sizeof(GenConsts)) == 0) return D3IL_OK;
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_gen_consts), &h->gc, sizeof(GenConsts)));
  HIPCHK(hipDeviceSynchronize());
  g_gen_loaded[h->device] = h->gc; g_gen_loaded_valid[h->device] = true;
  return D3IL_OK;
}
struct GenLaunch {      // scope = "constants checked ... kernel enqueued"
  std::unique_lock<std::mutex> lock;
  int rc;
  explicit GenLaunch(d3il_handle_s* h) : lock(g_model_mutex), rc(sync_gen_consts_locked(h)) {}
};

// the contact solvers' stopping rule lives in one __constant__ object per device; a handle whose setting differs from what is
// loaded re-loads it on its stream before launching (handles with different settings must not run concurrently on one device)
static int sync_solver_tol(d3il_handle_s* h, hipStream_t s) {
  static const SolverTol k_tol[2] = {SOLVER_TOL_PRODUCTION, SOLVER_TOL_STRICT};
  if (h->task_id == D3IL_TASK_AVOIDING) return D3IL_OK;
  std::lock_guard<std::mutex> lock(g_model_mutex);
  if (g_active_tol[h->device] == h->tol_mode) return D3IL_OK;
  // a switch of the rule is rare (parity A/B): fence the whole device on both sides of the copy, so that kernels of OTHER streams /
  // handles neither run across the change nor start before the copy has landed
  (void)s;
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_solver_tol), &k_tol[h->tol_mode ? 1 : 0], sizeof(SolverTol)));
  HIPCHK(hipDeviceSynchronize());
  g_active_tol[h->device] = h->tol_mode;
  return D3IL_OK;
}

// The fused rollout tail (k_avoiding_tail) has already drawn the NEXT step's action and advanced the harness pose by it.  Any other call that reads or writes
// the pose / the actions ends that sequence: the pose goes back to where the draw started (on the stream of the sequence, ordered before whatever follows on
// that stream; a caller that switches streams between calls synchronises them itself, as everywhere in this interface).
static int drop_prepared_action(d3il_handle_s* h) {
  if (!h->prep_valid) return D3IL_OK;
  h->prep_valid = false;
  HIPCHK(hipSetDevice(h->device));
  hipLaunchKernelGGL(k_restore_des, dim3((h->n + 255) / 256), dim3(256), 0, h->prep_stream, h->buf.policy_des, h->d_des_before, h->n, h->stride);
  HIPCHK(hipGetLastError());
  return D3IL_OK;
}
int d3il_reset(d3il_handle h, const uint8_t* env_mask, const double* contexts, void* stream) {
  if (!h) return fail(D3IL_EINVAL, "d3il_reset: null handle");
  if (int rc_ = drop_prepared_action(h)) return rc_;
  if (!h->started) return fail(D3IL_ESTATE, "d3il_reset: d3il_start() has not been called (env.start() before env.reset())");
  HIPCHK(hipSetDevice(h->device));
  d3il_buffers& b = h->buf;
  if (h->ctx_dim && !contexts) return fail(D3IL_EINVAL, "d3il_reset: this task needs contexts (device f64 [n_envs][ctx_dim]: Pushing 14, Sorting 7 n_boxes, Stacking 21, Aligning 14)");      // before anything is enqueued
  if (!h->ctx_dim && contexts) return fail(D3IL_EUNSUPPORTED, "d3il_reset: the Avoiding task takes no contexts");
  if (int rc = sync_solver_tol(h, (hipStream_t)stream)) return rc;
  if (h->ctx_dim && contexts && contexts != h->d_ctx) {
    int tot = h->n * h->ctx_dim;
    hipLaunchKernelGGL(k_store_contexts, dim3((tot + 255) / 256), dim3(256), 0, (hipStream_t)stream, env_mask, contexts, h->d_ctx, h->n, h->ctx_dim);
    HIPCHK(hipGetLastError());
  }
  if (gen_task(h->task_id)) {
    if (!contexts) return fail(D3IL_EINVAL, "d3il_reset: the task needs contexts (device f64 [n_envs][7 * n_boxes])");
    GenLaunch guard(h);
    if (guard.rc) return guard.rc;
    hipLaunchKernelGGL(k_sorting_reset, dim3((h->n + GEN_LANES - 1) / GEN_LANES), dim3(WAVE), GEN_LDS_H, (hipStream_t)stream, h->d_init_qpos, env_mask, contexts, b.state,
                       b.flags, b.step_count, b.obs, b.done, b.success, b.mode, h->d_scratch, h->n, h->stride);
    HIPCHK(hipGetLastError());
    return D3IL_OK;
  }
  if (h->task_id == D3IL_TASK_STACKING) {
    if (!contexts) return fail(D3IL_EINVAL, "d3il_reset: the Stacking task needs contexts (device f64 [n_envs][21])");
    StackLaunch guard(h);
    if (guard.rc) return guard.rc;
    if (h->stack_reset_coop)      // the step kernel in reset mode: cooperative phases, workgroups without a masked environment leave at once
      hipLaunchKernelGGL(k_stacking_step, dim3((h->n + SK_LANES - 1) / SK_LANES), dim3(WAVE), STACK_LDS, (hipStream_t)stream, b.state, b.flags, b.step_count, (const double*)nullptr, b.obs,
                         b.done, b.success, b.mode, b.info_f64, h->d_scratch, h->n, h->stride, 1, h->hc.max_steps, 1, env_mask, h->d_init_qpos, contexts);
    else
      hipLaunchKernelGGL(k_stacking_reset, dim3((h->n + SK_LANES - 1) / SK_LANES), dim3(WAVE), STACK_LDS_RESET, (hipStream_t)stream, h->d_init_qpos, env_mask, contexts, b.state,
                         b.flags, b.step_count, b.obs, b.done, b.success, b.mode, b.info_f64, h->d_scratch, h->n, h->stride);
    HIPCHK(hipGetLastError());
    return D3IL_OK;
  }
  if (h->task_id == D3IL_TASK_ALIGNING) {
    if (!contexts) return fail(D3IL_EINVAL, "d3il_reset: the Aligning task needs contexts (device f64 [n_envs][14]: box pos3 quat4 | target pos3 quat4)");
    StackLaunch guard(h);
    if (guard.rc) return guard.rc;
    hipLaunchKernelGGL(k_aligning_step, dim3((h->n + SK_LANES - 1) / SK_LANES), dim3(WAVE), STACK_LDS, (hipStream_t)stream, b.state, b.flags, b.step_count, (const double*)nullptr, b.obs,
                       b.done, b.success, b.mode, b.info_f64, h->d_scratch, h->n, h->stride, 1, h->hc.max_steps, 1, env_mask, h->d_init_qpos, contexts);
    HIPCHK(hipGetLastError());
    return D3IL_OK;
  }
  if (contexts) return fail(D3IL_EUNSUPPORTED, "d3il_reset: the Avoiding task takes no contexts");
  hipLaunchKernelGGL(k_avoiding_reset, dim3(h->stride / WAVE), dim3(WAVE), 0, (hipStream_t)stream, h->dc, h->d_init_qpos, env_mask, b.state, b.flags,
                     b.step_count, b.obs, b.done, b.success, b.mode, h->n, h->stride);
  HIPCHK(hipGetLastError());
  return D3IL_OK;
}

static int timing_drain_one(d3il_handle h) {
  const int slot = (int)(h->ring_drained % 128);
  float ms = 0;
  HIPCHK(hipEventSynchronize(h->ring1[slot]));
  HIPCHK(hipEventElapsedTime(&ms, h->ring0[slot], h->ring1[slot]));
  h->t_sum += ms; h->t_n++; if (ms < h->t_min) h->t_min = ms; if (ms > h->t_max) h->t_max = ms;
  h->ring_drained++;
  return D3IL_OK;
}
static int timing_begin(d3il_handle h, hipStream_t s) {
  if (h->ring_head - h->ring_drained >= 128) { if (int rc = timing_drain_one(h)) return rc; }      // the pair recorded 128 launches ago: long finished
  const int slot = (int)(h->ring_head % 128);
  h->ev0 = h->ring0[slot]; h->ev1 = h->ring1[slot];
  HIPCHK(hipEventRecord(h->ev0, s));
  return D3IL_OK;
}
int d3il_step(d3il_handle h, const double* actions, void* stream) {
  if (!h || !actions) return fail(D3IL_EINVAL, "d3il_step: null argument");
  if (int rc_ = drop_prepared_action(h)) return rc_;
  if (!h->started) return fail(D3IL_ESTATE, "d3il_step: d3il_start() has not been called");
  HIPCHK(hipSetDevice(h->device));
  d3il_buffers& b = h->buf;
  hipStream_t s = (hipStream_t)stream;
  if (int rc = sync_solver_tol(h, s)) return rc;
  if (gen_task(h->task_id)) {
    int nwgs = (h->n + GEN_LANES - 1) / GEN_LANES;
    GenLaunch guard(h);
    if (guard.rc) return guard.rc;
    if (h->timing) { if (int rc_ = timing_begin(h, s)) return rc_; }
    // the engine with contacts of the arm block only for models that evaluate rod <-> static box pairs (Inserting): the Sorting scenes run the
    // instantiation without that code
#define D3IL_GEN_LAUNCH(F, R) hipLaunchKernelGGL((k_sorting_step<F, R>), dim3(nwgs), dim3((1 + GEN_NSUB) * WAVE), GEN_LDS_STEP, s, b.state, b.flags, b.step_count, actions, b.obs, b.done, \
                                                  b.success, b.mode, h->d_scratch, h->n, h->stride, h->hc.n_substeps, h->hc.max_steps)
    if (h->gc.rod_static) { if (h->fast) D3IL_GEN_LAUNCH(true, true); else D3IL_GEN_LAUNCH(false, true); }
    else { if (h->fast) D3IL_GEN_LAUNCH(true, false); else D3IL_GEN_LAUNCH(false, false); }
#undef D3IL_GEN_LAUNCH
    HIPCHK(hipGetLastError());
    if (h->timing) { HIPCHK(hipEventRecord(h->ev1, s)); h->ev_valid = true; h->ring_head++; }
    return D3IL_OK;
  }
  if (h->task_id == D3IL_TASK_STACKING) {
    StackLaunch guard(h);
    if (guard.rc) return guard.rc;
    if (h->timing) { if (int rc_ = timing_begin(h, s)) return rc_; }
    hipLaunchKernelGGL(k_stacking_step, dim3((h->n + SK_LANES - 1) / SK_LANES), dim3(WAVE), STACK_LDS, s, b.state, b.flags, b.step_count, actions, b.obs, b.done, b.success, b.mode,
                       b.info_f64, h->d_scratch, h->n, h->stride, h->hc.n_substeps, h->hc.max_steps, 0, (const unsigned char*)nullptr, (const double*)nullptr, (const double*)nullptr);
    HIPCHK(hipGetLastError());
    if (h->timing) { HIPCHK(hipEventRecord(h->ev1, s)); h->ev_valid = true; h->ring_head++; }
    return D3IL_OK;
  }
  if (h->task_id == D3IL_TASK_ALIGNING) {
    StackLaunch guard(h);
    if (guard.rc) return guard.rc;
    if (h->timing) { if (int rc_ = timing_begin(h, s)) return rc_; }
    hipLaunchKernelGGL(k_aligning_step, dim3((h->n + SK_LANES - 1) / SK_LANES), dim3(WAVE), STACK_LDS, s, b.state, b.flags, b.step_count, actions, b.obs, b.done, b.success, b.mode,
                       b.info_f64, h->d_scratch, h->n, h->stride, h->hc.n_substeps, h->hc.max_steps, 0, (const unsigned char*)nullptr, (const double*)nullptr, (const double*)nullptr);
    HIPCHK(hipGetLastError());
    if (h->timing) { HIPCHK(hipEventRecord(h->ev1, s)); h->ev_valid = true; h->ring_head++; }
    return D3IL_OK;
  }
  // Workgroup placement: a workgroup is one wave; the dispatcher packs several of them onto one CU (and SIMD) before
  // moving on, which halves the per-wave issue rate when only a few hundred waves exist.  Requesting LDS that is not
  // otherwise needed caps the workgroups per CU so that the waves spread over all 256 CUs / 1024 SIMDs.
  int nwg = (h->n + h->lanes - 1) / h->lanes, lds = h->lds_pad;
  if (lds < 0) {
    int per_cu = (nwg + 255) / 256;                       // workgroups each CU has to host
    lds = per_cu >= 8 ? 0 : (160 * 1024 / per_cu) - 1024;  // leave slack below the 160 KiB per-CU pool
    if (lds > 64 * 1024) lds = 64 * 1024;
  }
  if (h->timing && !h->rg_capturing) { if (int rc_ = timing_begin(h, s)) return rc_; }
  // the two-wave kernel wins at every batch size measured (4096 ... 262144 envs: +64 % ... +20 %): at small N the second
  // wave uses an idle SIMD, at saturation its 256-VGPR roles run two waves per SIMD and hide FP64 latency
  bool split = h->fast && h->lanes == WAVE && h->split != 0;
  // the third wave (rare constraint paths) while a CU hosts one workgroup anyway; at saturation the two-wave form keeps two workgroups per CU
  if (split && nwg <= h->serve_max_wg)
    hipLaunchKernelGGL((k_avoiding_step_split<true, true>), dim3(nwg), dim3(3 * WAVE), AVOID_LDS_SERVE, s, h->dc, b.state, b.flags, b.step_count, actions, b.obs, b.done,
                       b.success, b.mode, h->n, h->stride, h->hc.n_substeps, h->hc.max_steps);
  else if (split)
    hipLaunchKernelGGL((k_avoiding_step_split<true, false>), dim3(nwg), dim3(2 * WAVE), 0, s, h->dc, b.state, b.flags, b.step_count, actions, b.obs, b.done,
                       b.success, b.mode, h->n, h->stride, h->hc.n_substeps, h->hc.max_steps);
  else if (h->fast)
    hipLaunchKernelGGL((k_avoiding_step<true, true>), dim3(nwg), dim3(WAVE), lds, s, h->dc, b.state, b.flags, b.step_count, actions, b.obs, b.done,
                       b.success, b.mode, h->n, h->stride, h->hc.n_substeps, h->hc.max_steps, h->lanes);
  else
    hipLaunchKernelGGL((k_avoiding_step<false, true>), dim3(nwg), dim3(WAVE), lds, s, h->dc, b.state, b.flags, b.step_count, actions, b.obs, b.done,
                       b.success, b.mode, h->n, h->stride, h->hc.n_substeps, h->hc.max_steps, h->lanes);
  HIPCHK(hipGetLastError());
  if (h->timing && !h->rg_capturing) { HIPCHK(hipEventRecord(h->ev1, s)); h->ev_valid = true; h->ring_head++; }
  return D3IL_OK;
}

int d3il_get_buffers(d3il_handle h, d3il_buffers* out) {
  if (!h || !out) return fail(D3IL_EINVAL, "d3il_get_buffers: null argument");
  *out = h->buf;
  return D3IL_OK;
}

static int check_state_rows(d3il_handle h, const void* state, int32_t state_rows, const char* who) {
  if (state && state_rows != h->state_rows)
    return fail(D3IL_EINVAL, std::string(who) + ": the caller's buffer has " + std::to_string(state_rows) + " state rows, this handle has " + std::to_string(h->state_rows) +
                                 " (d3il_buffers.state_rows)");
  return D3IL_OK;
}
int d3il_get_state(d3il_handle h, double* state, int32_t state_rows, uint32_t* flags, int32_t* steps) {
  if (!h) return fail(D3IL_EINVAL, "d3il_get_state: null handle");
  if (int rc_ = check_state_rows(h, state, state_rows, "d3il_get_state")) return rc_;
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipDeviceSynchronize());
  if (state) HIPCHK(hipMemcpy2D(state, (size_t)h->n * 8, h->buf.state, (size_t)h->stride * 8, (size_t)h->n * 8, h->state_rows, hipMemcpyDeviceToHost));
  if (flags) HIPCHK(hipMemcpy(flags, h->buf.flags, (size_t)h->n * 4, hipMemcpyDeviceToHost));
  if (steps) HIPCHK(hipMemcpy(steps, h->buf.step_count, (size_t)h->n * 4, hipMemcpyDeviceToHost));
  return D3IL_OK;
}
int d3il_set_state(d3il_handle h, const double* state, int32_t state_rows, const uint32_t* flags, const int32_t* steps) {
  if (!h) return fail(D3IL_EINVAL, "d3il_set_state: null handle");
  if (int rc_ = check_state_rows(h, state, state_rows, "d3il_set_state")) return rc_;
  if (int rc_ = drop_prepared_action(h)) return rc_;
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipDeviceSynchronize());
  if (state) HIPCHK(hipMemcpy2D(h->buf.state, (size_t)h->stride * 8, state, (size_t)h->n * 8, (size_t)h->n * 8, h->state_rows, hipMemcpyHostToDevice));
  if (flags) HIPCHK(hipMemcpy(h->buf.flags, flags, (size_t)h->n * 4, hipMemcpyHostToDevice));
  if (steps) HIPCHK(hipMemcpy(h->buf.step_count, steps, (size_t)h->n * 4, hipMemcpyHostToDevice));
  return D3IL_OK;
}

int d3il_policy_begin(d3il_handle h, const uint8_t* env_mask, void* stream) {
  if (!h) return fail(D3IL_EINVAL, "d3il_policy_begin: null handle");
  if (int rc_ = drop_prepared_action(h)) return rc_;
  HIPCHK(hipSetDevice(h->device));
  hipLaunchKernelGGL(k_policy_begin, dim3((h->n + 255) / 256), dim3(256), 0, (hipStream_t)stream, env_mask, h->buf.state, h->buf.policy_des, h->n, h->stride);
  HIPCHK(hipGetLastError());
  return D3IL_OK;
}
int d3il_policy_action(d3il_handle h, uint64_t seed, uint64_t env_offset, uint32_t t, double* actions, void* stream) {
  if (!h || !actions) return fail(D3IL_EINVAL, "d3il_policy_action: null argument");
  if (int rc_ = drop_prepared_action(h)) return rc_;
  if (h->buf.action_dim != 7) return fail(D3IL_EUNSUPPORTED, "d3il_policy_action: the random Cartesian policy writes 7-wide rows (Avoiding / Pushing / Sorting); Stacking actions are 8 wide");
  HIPCHK(hipSetDevice(h->device));
  hipLaunchKernelGGL(k_policy_action, dim3((h->n + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->buf.policy_des, actions, (unsigned long long)seed,
                     (unsigned long long)env_offset, t, h->n, h->stride);
  HIPCHK(hipGetLastError());
  return D3IL_OK;
}
int d3il_attention_causal_f32(const float* qkv, float* out, int B, int T, int H, int D, void* stream) {
  if (!qkv || !out) return fail(D3IL_EINVAL, "d3il_attention_causal_f32: null argument");
  if (B < 0 || T < 1 || T > 32 || H < 1 || D < 1 || D > 32) return fail(D3IL_EINVAL, "d3il_attention_causal_f32: needs 1 <= T <= 32, 1 <= D <= 32");
  const long total = (long)B * H * ((T + 1) / 2);      // one lane per pair of queries (i, T - 1 - i)
  if (total == 0) return D3IL_OK;
  const dim3 grid((unsigned)((total + 255) / 256)), block(256);
  const bool aligned = D % 4 == 0 && ((uintptr_t)qkv % 16) == 0 && ((uintptr_t)out % 16) == 0;
  if (aligned && D == 20) hipLaunchKernelGGL(k_attention_causal_f32<5>, grid, block, 0, (hipStream_t)stream, qkv, out, B, T, H, D);
  else if (aligned && D == 16) hipLaunchKernelGGL(k_attention_causal_f32<4>, grid, block, 0, (hipStream_t)stream, qkv, out, B, T, H, D);
  else if (aligned && D == 32) hipLaunchKernelGGL(k_attention_causal_f32<8>, grid, block, 0, (hipStream_t)stream, qkv, out, B, T, H, D);
  else hipLaunchKernelGGL(k_attention_causal_f32<0>, grid, block, 0, (hipStream_t)stream, qkv, out, B, T, H, D);
  HIPCHK(hipGetLastError());
  return D3IL_OK;
}
int d3il_layernorm_f32(const float* x, const float* weight, const float* bias, float* y, long rows, int C, float eps, void* stream) {
  if (!x || !weight || !bias || !y) return fail(D3IL_EINVAL, "d3il_layernorm_f32: null argument");
  if (rows < 0 || C < 4 || C > 128 || C % 4 != 0) return fail(D3IL_EINVAL, "d3il_layernorm_f32: needs 4 <= C <= 128, C a multiple of 4");
  if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)weight | (uintptr_t)bias) % 16 != 0) return fail(D3IL_EINVAL, "d3il_layernorm_f32: pointers must be 16-byte aligned");
  if (rows == 0) return D3IL_OK;
  const long threads = rows * 32;
  hipLaunchKernelGGL(k_layernorm_f32, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, weight, bias, y, rows, C, eps);
  HIPCHK(hipGetLastError());
  return D3IL_OK;
}
int d3il_linear120_f32(const float* xin, const float* ln_weight, const float* ln_bias, float ln_eps, const float* w_packed, const float* bias, const float* resid, float* out,
                       long rows, int N, void* stream) {
  if (!xin || !w_packed || !bias || !out) return fail(D3IL_EINVAL, "d3il_linear120_f32: null argument");
  if ((ln_weight == nullptr) != (ln_bias == nullptr)) return fail(D3IL_EINVAL, "d3il_linear120_f32: LayerNorm weight and bias come together");
  if (rows < 0 || N < 4 || N % 4 != 0) return fail(D3IL_EINVAL, "d3il_linear120_f32: needs rows >= 0 and N a positive multiple of 4");
  if (((uintptr_t)xin | (uintptr_t)w_packed | (uintptr_t)bias | (uintptr_t)resid | (uintptr_t)out) % 16 != 0) return fail(D3IL_EINVAL, "d3il_linear120_f32: pointers must be 16-byte aligned");
  if (rows == 0) return D3IL_OK;
  hipLaunchKernelGGL(k_linear120_f32, dim3((unsigned)((rows + 63) / 64)), dim3(256), 0, (hipStream_t)stream, xin, w_packed, bias, resid, out, rows, N, ln_weight, ln_bias, ln_eps);
  HIPCHK(hipGetLastError());
  return D3IL_OK;
}
int d3il_ddpm_mlp_f32(const float* state, const float* noise, const float* temb, const float* w_in, const float* b_in, const float* w_blocks, const float* b_blocks,
                      const float* w_out, const float* b_out, const float* sched, const float* bounds, float* out, long rows, int state_dim, int n_timesteps, int hidden,
                      int n_blocks, void* stream) {
  if (!state || !noise || !temb || !w_in || !b_in || !w_blocks || !b_blocks || !w_out || !b_out || !sched || !bounds || !out) return fail(D3IL_EINVAL, "d3il_ddpm_mlp_f32: null argument");
  if (hidden != DD_H) return fail(D3IL_EUNSUPPORTED, "d3il_ddpm_mlp_f32: built for hidden 256 (the DiffusionMLP of the DDPM configs)");
  if (state_dim < 1 || state_dim > 18) return fail(D3IL_EUNSUPPORTED, "d3il_ddpm_mlp_f32: action 2 + time embedding 8 + state must fit 28 inputs (state_dim <= 18)");
  if (n_timesteps < 1 || n_blocks < 0 || rows < 0) return fail(D3IL_EINVAL, "d3il_ddpm_mlp_f32: bad counts");
  if (((uintptr_t)w_in | (uintptr_t)b_in | (uintptr_t)w_blocks | (uintptr_t)b_blocks | (uintptr_t)w_out) % 16 != 0) return fail(D3IL_EINVAL, "d3il_ddpm_mlp_f32: weights and biases must be 16-byte aligned");
  if (rows == 0) return D3IL_OK;
  hipLaunchKernelGGL(k_ddpm_mlp_f32, dim3((unsigned)((rows + 15) / 16)), dim3(64 * DD_NW), 0, (hipStream_t)stream, state, noise, temb, w_in, b_in, w_blocks, b_blocks, w_out, b_out, sched, bounds, out,
                     rows, state_dim, n_timesteps, n_blocks);
  HIPCHK(hipGetLastError());
  return D3IL_OK;
}
int d3il_resmlp_f32(const float* x, const float* w_in, const float* b_in, const float* w_blocks, const float* b_blocks, const float* w_out, const float* b_out, float* out, long rows,
                    int in_dim, int hidden, int n_blocks, int out_dim, void* stream) {
  if (!x || !w_in || !b_in || !w_blocks || !b_blocks || !w_out || !b_out || !out) return fail(D3IL_EINVAL, "d3il_resmlp_f32: null argument");
  if (hidden != 128 && hidden != 256) return fail(D3IL_EUNSUPPORTED, "d3il_resmlp_f32: built for hidden 128 and 256 (the ResidualMLPNetwork of the BC and DDPM configs)");
  if (in_dim < 1 || in_dim > 28 || out_dim < 1 || out_dim > 16) return fail(D3IL_EUNSUPPORTED, "d3il_resmlp_f32: at most 28 inputs and 16 outputs");
  if (n_blocks < 0 || rows < 0) return fail(D3IL_EINVAL, "d3il_resmlp_f32: bad counts");
  if (((uintptr_t)w_in | (uintptr_t)b_in | (uintptr_t)w_blocks | (uintptr_t)b_blocks | (uintptr_t)w_out) % 16 != 0) return fail(D3IL_EINVAL, "d3il_resmlp_f32: weights and biases must be 16-byte aligned");
  if (rows == 0) return D3IL_OK;
  const dim3 grid((unsigned)((rows + 15) / 16)), block(512);
  if (hidden == 128) hipLaunchKernelGGL(k_resmlp_f32<128>, grid, block, 0, (hipStream_t)stream, x, w_in, b_in, w_blocks, b_blocks, w_out, b_out, out, rows, in_dim, out_dim, n_blocks);
  else hipLaunchKernelGGL(k_resmlp_f32<256>, grid, block, 0, (hipStream_t)stream, x, w_in, b_in, w_blocks, b_blocks, w_out, b_out, out, rows, in_dim, out_dim, n_blocks);
  HIPCHK(hipGetLastError());
  return D3IL_OK;
}
int d3il_mlp_gelu_residual_f32(const float* h, const float* x, const float* w_packed, const float* b1, const float* b2, float* out, long rows, int C, int H, void* stream) {
  return d3il_mlp_ln_gelu_residual_f32(h, nullptr, nullptr, 0.f, x, w_packed, b1, b2, out, rows, C, H, stream);
}
int d3il_mlp_ln_gelu_residual_f32(const float* h, const float* ln_weight, const float* ln_bias, float ln_eps, const float* x, const float* w_packed, const float* b1, const float* b2,
                                  float* out, long rows, int C, int H, void* stream) {
  if ((ln_weight == nullptr) != (ln_bias == nullptr)) return fail(D3IL_EINVAL, "d3il_mlp_ln_gelu_residual_f32: LayerNorm weight and bias come together");
  if (!h || !x || !w_packed || !b1 || !b2 || !out) return fail(D3IL_EINVAL, "d3il_mlp_gelu_residual_f32: null argument");
  if (C != MLP_C || H != MLP_H) return fail(D3IL_EUNSUPPORTED, "d3il_mlp_gelu_residual_f32: built for n_embd 120, hidden 480 (the DiffusionGPT of the BESO configs)");
  if (rows < 0) return fail(D3IL_EINVAL, "d3il_mlp_gelu_residual_f32: negative row count");
  if (((uintptr_t)h | (uintptr_t)x | (uintptr_t)w_packed | (uintptr_t)b2 | (uintptr_t)out) % 16 != 0) return fail(D3IL_EINVAL, "d3il_mlp_gelu_residual_f32: pointers must be 16-byte aligned");
  if (rows == 0) return D3IL_OK;
  hipLaunchKernelGGL(k_mlp_gelu_residual_f32, dim3((unsigned)((rows + 63) / 64)), dim3(256), 0, (hipStream_t)stream, h, x, w_packed, b1, b2, out, rows, ln_weight, ln_bias, ln_eps);
  HIPCHK(hipGetLastError());
  return D3IL_OK;
}
int d3il_mlp_ln_gelu_residual_f16x3(const float* h, const float* ln_weight, const float* ln_bias, float ln_eps, const float* x, const void* w_packed, const float* b1, const float* b2,
                                    float* out, long rows, int C, int H, void* stream) {
  if ((ln_weight == nullptr) != (ln_bias == nullptr)) return fail(D3IL_EINVAL, "d3il_mlp_ln_gelu_residual_f16x3: LayerNorm weight and bias come together");
  if (!h || !x || !w_packed || !b1 || !b2 || !out) return fail(D3IL_EINVAL, "d3il_mlp_ln_gelu_residual_f16x3: null argument");
  if (C != HX_C || H != HX_H) return fail(D3IL_EUNSUPPORTED, "d3il_mlp_ln_gelu_residual_f16x3: built for n_embd 120, hidden 480 (the DiffusionGPT of the BESO configs)");
  if (rows < 0) return fail(D3IL_EINVAL, "d3il_mlp_ln_gelu_residual_f16x3: negative row count");
  if (((uintptr_t)h | (uintptr_t)x | (uintptr_t)w_packed | (uintptr_t)b2 | (uintptr_t)out) % 16 != 0) return fail(D3IL_EINVAL, "d3il_mlp_ln_gelu_residual_f16x3: pointers must be 16-byte aligned");
  if (rows == 0) return D3IL_OK;
  hipLaunchKernelGGL(k_mlp_gelu_residual_f16x3<HX_MLP_NW>, dim3((unsigned)((rows + 16 * HX_MLP_NW - 1) / (16 * HX_MLP_NW))), dim3(64 * HX_MLP_NW), 0, (hipStream_t)stream, h, x, (const hx_h8*)w_packed, b1, b2, out, rows,
                     ln_weight, ln_bias, ln_eps);
  HIPCHK(hipGetLastError());
  return D3IL_OK;
}
int d3il_linear120_f16x3(const float* xin, const float* ln_weight, const float* ln_bias, float ln_eps, const void* w_packed, const float* bias, const float* resid, float* out,
                         long rows, int N, void* stream) {
  if (!xin || !w_packed || !bias || !out) return fail(D3IL_EINVAL, "d3il_linear120_f16x3: null argument");
  if ((ln_weight == nullptr) != (ln_bias == nullptr)) return fail(D3IL_EINVAL, "d3il_linear120_f16x3: LayerNorm weight and bias come together");
  if (rows < 0 || N < 4 || N % 4 != 0 || N > 384) return fail(D3IL_EINVAL, "d3il_linear120_f16x3: needs rows >= 0 and N a multiple of 4 in 4 .. 384");
  if (((uintptr_t)xin | (uintptr_t)w_packed | (uintptr_t)bias | (uintptr_t)resid | (uintptr_t)out) % 16 != 0) return fail(D3IL_EINVAL, "d3il_linear120_f16x3: pointers must be 16-byte aligned");
  if (rows == 0) return D3IL_OK;
  hipLaunchKernelGGL(k_linear120_f16x3<HX_LIN_NW>, dim3((unsigned)((rows + 16 * HX_LIN_NW - 1) / (16 * HX_LIN_NW))), dim3(64 * HX_LIN_NW), 0, (hipStream_t)stream, xin, (const hx_h8*)w_packed, bias, resid, out, rows, N,
                     ln_weight, ln_bias, ln_eps);
  HIPCHK(hipGetLastError());
  return D3IL_OK;
}
int d3il_attn_half_f16x3(const float* x, const float* ln_weight, const float* ln_bias, float ln_eps, const void* w_packed, const float* b_qkv, const float* b_proj, float* out,
                         long n_seq, int T, int n_head, int C, void* stream) {
  if (!x || !ln_weight || !ln_bias || !w_packed || !b_qkv || !b_proj || !out) return fail(D3IL_EINVAL, "d3il_attn_half_f16x3: null argument");
  if (C != HX_C || n_head != 6) return fail(D3IL_EUNSUPPORTED, "d3il_attn_half_f16x3: built for 120 features in 6 heads (the DiffusionGPT of the BESO configs)");
  if (n_seq < 0 || T < 1 || T > 16) return fail(D3IL_EINVAL, "d3il_attn_half_f16x3: needs n_seq >= 0 and 1 <= T <= 16 (one sequence per matrix-core tile)");
  if (((uintptr_t)x | (uintptr_t)w_packed | (uintptr_t)b_qkv | (uintptr_t)b_proj | (uintptr_t)out | (uintptr_t)ln_weight | (uintptr_t)ln_bias) % 16 != 0) return fail(D3IL_EINVAL, "d3il_attn_half_f16x3: pointers must be 16-byte aligned");
  if (x == out) return fail(D3IL_EINVAL, "d3il_attn_half_f16x3: out must not alias x (the residual is read after other rows have been written)");
  if (n_seq == 0) return D3IL_OK;
  if (T <= 11)
    hipLaunchKernelGGL((k_attn_half_f16x3<8, 11>), dim3((unsigned)((n_seq + 7) / 8)), dim3(512), 0, (hipStream_t)stream, x, (const hx_h8*)w_packed, b_qkv, b_proj, out, n_seq, T, ln_weight, ln_bias, ln_eps);
  else
    hipLaunchKernelGGL((k_attn_half_f16x3<4, 16>), dim3((unsigned)((n_seq + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, (const hx_h8*)w_packed, b_qkv, b_proj, out, n_seq, T, ln_weight, ln_bias, ln_eps);
  HIPCHK(hipGetLastError());
  return D3IL_OK;
}
int d3il_set_tally(d3il_handle h, const int32_t* ctx_id_device, int n_ctx, int64_t* table_device) {
  if (!h) return fail(D3IL_EINVAL, "d3il_set_tally: null handle");
  if (table_device && n_ctx <= 0) return fail(D3IL_EINVAL, "d3il_set_tally: n_ctx must be positive");
  if (h->rg_ready) { HIPCHK(hipDeviceSynchronize()); rg_drop(h); }
  h->tally_ctx = ctx_id_device; h->tally_nctx = n_ctx; h->tally_table = table_device;
  return D3IL_OK;
}

int d3il_auto_reset(d3il_handle h, int64_t* episode_counts_device, void* stream) {
  if (!h) return fail(D3IL_EINVAL, "d3il_auto_reset: null handle");
  if (int rc_ = drop_prepared_action(h)) return rc_;
  if (!h->started) return fail(D3IL_ESTATE, "d3il_auto_reset: d3il_start() has not been called");
  HIPCHK(hipSetDevice(h->device));
  d3il_buffers& b = h->buf;
  hipStream_t s = (hipStream_t)stream;
  const bool avoiding = h->task_id == D3IL_TASK_AVOIDING;
  if (avoiding && !episode_counts_device) return fail(D3IL_EINVAL, "d3il_auto_reset: the Avoiding task needs episode_counts (device i64[2])");   // before anything is enqueued
  // which environments this call resets (buf.last_reset): the harness re-latches its per-lane state (agent history) from it
  HIPCHK(hipMemcpyAsync(h->d_mask, b.done, (size_t)h->n, hipMemcpyDeviceToDevice, s));
  if (h->tally_table || (!avoiding && episode_counts_device)) {
    // Avoiding counts its episodes in the fused reset kernel below
    hipLaunchKernelGGL(k_episode_tally, dim3((h->n + 255) / 256), dim3(256), 0, s, b.done, b.success, b.mode, h->tally_ctx, (long long*)h->tally_table,
                       avoiding ? (long long*)nullptr : (long long*)episode_counts_device, h->n, h->tally_nctx, (h->task_id == D3IL_TASK_PUSHING || h->task_id == D3IL_TASK_ALIGNING) ? 1 : 0,
                       (h->task_id == D3IL_TASK_STACKING || h->task_id == D3IL_TASK_INSERTING) ? 1 : 0);
    HIPCHK(hipGetLastError());
  }
  if (avoiding) {
    hipLaunchKernelGGL(k_avoiding_auto_reset, dim3(h->stride / WAVE), dim3(WAVE), 0, s, h->dc, h->d_init_qpos, b.state, b.flags, b.step_count,
                       b.obs, b.done, b.success, b.mode, b.policy_des, (long long*)episode_counts_device, h->n, h->stride);
    HIPCHK(hipGetLastError());
    return D3IL_OK;
  }
  // Pushing / Sorting: the next trajectory of a lane starts from the context of its last reset (pushing_sim.py:63, sorting_sim.py:112)
  if (int rc = d3il_reset(h, h->d_mask, h->d_ctx, stream)) return rc;
  return d3il_policy_begin(h, h->d_mask, stream);
}

int d3il_count_metrics(d3il_handle h, int64_t* out_counts_device, void* stream) {
  if (!h || !out_counts_device) return fail(D3IL_EINVAL, "d3il_count_metrics: null argument");
  if (h->task_id != D3IL_TASK_AVOIDING) return fail(D3IL_EUNSUPPORTED, "d3il_count_metrics: Avoiding only");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipMemsetAsync(out_counts_device, 0, (2 + 512) * sizeof(int64_t), (hipStream_t)stream));
  hipLaunchKernelGGL(k_count_metrics, dim3((h->n + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->buf.done, h->buf.flags, (long long*)out_counts_device, h->n);
  HIPCHK(hipGetLastError());
  return D3IL_OK;
}
// ---- cross-GPU reduction of the integer metric tables with RCCL, inside the library (north_star: "RCCL ... over xGMI only
// for the final success-rate / entropy reduction").  RCCL is resolved at run time: the symbols of an RCCL already loaded into the process
// (PyTorch-ROCm ships one: a communicator must be used with the library that made it) or, failing that, librccl.so from the ROCm install;
// the library itself links only the HIP runtime, so single-GPU users need no RCCL at all.
namespace {
struct RcclApi {
  int (*GetUniqueId)(void*);
  int (*CommInitRank)(void**, int, d3il_rccl_unique_id, int);     // ncclUniqueId is a 128-byte struct passed by value
  int (*CommDestroy)(void*);
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
  int (*CommCount)(void*, int*);
  const char* (*GetErrorString)(int);
  bool ok;
};
RcclApi* rccl_api() {
  static RcclApi api = [] {
    RcclApi a{}; a.ok = false;
    void* hnd = RTLD_DEFAULT;
    if (!dlsym(hnd, "ncclAllReduce")) {
      hnd = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
      if (!hnd) hnd = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
      if (!hnd) hnd = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
      if (!hnd) return a;
    }
    a.GetUniqueId = (int (*)(void*))dlsym(hnd, "ncclGetUniqueId");
    a.CommInitRank = (int (*)(void**, int, d3il_rccl_unique_id, int))dlsym(hnd, "ncclCommInitRank");
    a.CommDestroy = (int (*)(void*))dlsym(hnd, "ncclCommDestroy");
    a.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(hnd, "ncclAllReduce");
    a.CommCount = (int (*)(void*, int*))dlsym(hnd, "ncclCommCount");
    a.GetErrorString = (const char* (*)(int))dlsym(hnd, "ncclGetErrorString");
    a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllReduce;
    return a;
  }();
  return &api;
}
int rccl_fail(const char* what, int rc) {
  RcclApi* a = rccl_api();
  return fail(D3IL_ERCCL, std::string(what) + ": " + (a->GetErrorString ? a->GetErrorString(rc) : "RCCL error") + " (" + std::to_string(rc) + ")");
}
}  // namespace

int d3il_rccl_available(void) { return rccl_api()->ok ? 1 : 0; }
int d3il_comm_count(d3il_comm comm, int* ranks) {
  if (!comm || !ranks) return fail(D3IL_EINVAL, "d3il_comm_count: null argument");
  RcclApi* a = rccl_api();
  if (!a->ok || !a->CommCount) return fail(D3IL_ERCCL, "d3il_comm_count: RCCL could not be resolved");
  if (int rc = a->CommCount((void*)comm, ranks)) return rccl_fail("ncclCommCount", rc);
  return D3IL_OK;
}
int d3il_comm_unique_id(d3il_rccl_unique_id* out) {
  if (!out) return fail(D3IL_EINVAL, "d3il_comm_unique_id: null argument");
  RcclApi* a = rccl_api();
  if (!a->ok) return fail(D3IL_ERCCL, "d3il_comm_unique_id: RCCL (librccl.so) could not be resolved");
  if (int rc = a->GetUniqueId(out)) return rccl_fail("ncclGetUniqueId", rc);
  return D3IL_OK;
}
int d3il_comm_init(const d3il_rccl_unique_id* id, int rank, int world, int device_id, d3il_comm* out) {
  if (!id || !out) return fail(D3IL_EINVAL, "d3il_comm_init: null argument");
  if (world < 1 || rank < 0 || rank >= world) return fail(D3IL_EINVAL, "d3il_comm_init: need 0 <= rank < world");
  RcclApi* a = rccl_api();
  if (!a->ok) return fail(D3IL_ERCCL, "d3il_comm_init: RCCL (librccl.so) could not be resolved");
  HIPCHK(hipSetDevice(device_id));
  void* comm = nullptr;
  if (int rc = a->CommInitRank(&comm, world, *id, rank)) return rccl_fail("ncclCommInitRank", rc);
  *out = (d3il_comm)comm;
  return D3IL_OK;
}
int d3il_comm_destroy(d3il_comm comm) {
  if (!comm) return fail(D3IL_EINVAL, "d3il_comm_destroy: null communicator");
  RcclApi* a = rccl_api();
  if (!a->ok) return fail(D3IL_ERCCL, "d3il_comm_destroy: RCCL could not be resolved");
  if (int rc = a->CommDestroy((void*)comm)) return rccl_fail("ncclCommDestroy", rc);
  return D3IL_OK;
}
int d3il_reduce_metrics(d3il_handle h, d3il_comm comm, int64_t* table_device, size_t count, void* stream) {
  if (!comm || (!h && !table_device)) return fail(D3IL_EINVAL, "d3il_reduce_metrics: null argument");      // h may be NULL with an explicit table (a rank without environments)
  if (!table_device) { table_device = h->tally_table; count = (size_t)h->tally_nctx * D3IL_TALLY_ROW; }      // default: the table of d3il_set_tally
  if (!table_device || count == 0) return fail(D3IL_EINVAL, "d3il_reduce_metrics: no table (pass one, or register it with d3il_set_tally)");
  RcclApi* a = rccl_api();
  if (!a->ok) return fail(D3IL_ERCCL, "d3il_reduce_metrics: RCCL could not be resolved");
  if (h) HIPCHK(hipSetDevice(h->device));
  // ONE in-place all-reduce(sum) of int64 counts: integer sums are bit-exact and independent of the rank order (ncclInt64 = 4, ncclSum = 0)
  if (int rc = a->AllReduce(table_device, table_device, count, 4, 0, (void*)comm, (hipStream_t)stream)) return rccl_fail("ncclAllReduce", rc);
  return D3IL_OK;
}

int d3il_set_timing(d3il_handle h, int enabled) {
  if (!h) return fail(D3IL_EINVAL, "d3il_set_timing: null handle");
  HIPCHK(hipSetDevice(h->device));
  if (enabled && !h->ring_created) {
    for (int i = 0; i < 128; i++) { HIPCHK(hipEventCreate(&h->ring0[i])); HIPCHK(hipEventCreate(&h->ring1[i])); }
    h->ring_created = true;
  }
  if (h->rg_ready) { HIPCHK(hipDeviceSynchronize()); rg_drop(h); }
  if (enabled) { HIPCHK(hipDeviceSynchronize()); h->ring_head = h->ring_drained = h->t_n = 0; h->t_sum = 0; h->t_min = 1e300; h->t_max = 0; }
  h->timing = enabled != 0; h->ev_valid = false;
  return D3IL_OK;
}
int d3il_timing_stats(d3il_handle h, double* out4) {
  if (!h || !out4) return fail(D3IL_EINVAL, "d3il_timing_stats: null argument");
  HIPCHK(hipSetDevice(h->device));
  while (h->ring_drained < h->ring_head) { if (int rc = timing_drain_one(h)) return rc; }
  out4[0] = h->t_sum; out4[1] = h->t_n ? h->t_min : 0.0; out4[2] = h->t_max; out4[3] = (double)h->t_n;
  return D3IL_OK;
}
int d3il_step_auto_reset(d3il_handle h, const double* actions, int64_t* episode_counts_device, void* stream) {
  if (int rc = d3il_step(h, actions, stream)) return rc;
  return d3il_auto_reset(h, episode_counts_device, stream);
}
// ---- the captured rollout step (option graph_rollout).  One step of the random-policy harness is eight runtime calls (policy kernel, two event records, step
// kernel, mask copy, tally kernel, auto-reset kernel) - ~85 us of host time, which bounds the number of sub-batches a process can keep in flight (S = 8:
// host bound).  Captured once per handle into RG_SLOTS graphs (own event pair each; the step counter in device memory, advanced by the graph's last node), a
// step is ONE hipGraphLaunch.  Anything that changes what a step launches (options, timing, tally, other arguments) drops the graphs; they are re-captured
// by the next call.
static void rg_drop(d3il_handle_s* h) {
  if (!h->rg_ready) return;
  for (int k = 0; k < RG_SLOTS; k++) { (void)hipGraphExecDestroy(h->rg_exec[k]); (void)hipGraphDestroy(h->rg_graph[k]); }
  h->rg_ready = false;
}
static int rg_capture(d3il_handle h, uint64_t seed, uint64_t env_offset, uint32_t t, double* actions, int64_t* counts, hipStream_t s) {
  // a prepared action left by a fused-tail step is dropped BEFORE the capture begins: inside it, d3il_step's own drop would record k_restore_des into the
  // graph and every replay would put a stale harness pose back
  if (int rc_ = drop_prepared_action(h)) return rc_;
  if (!h->rg_t_dev) HIPCHK(hipMalloc(&h->rg_t_dev, sizeof(unsigned)));
  HIPCHK(hipMemcpyAsync(h->rg_t_dev, &t, sizeof(unsigned), hipMemcpyHostToDevice, s));
  HIPCHK(hipStreamSynchronize(s));
  h->rg_timing = h->timing;
  for (int k = 0; k < RG_SLOTS; k++) {
    HIPCHK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
    h->rg_capturing = true; h->rg_slot = k;
    hipLaunchKernelGGL(k_policy_action_dev, dim3((h->n + 255) / 256), dim3(256), 0, s, h->buf.policy_des, actions, (unsigned long long)seed, (unsigned long long)env_offset,
                       (const unsigned*)h->rg_t_dev, h->n, h->stride);
    int rc = d3il_step(h, actions, s);
    if (!rc) rc = d3il_auto_reset(h, counts, s);
    if (!rc) hipLaunchKernelGGL(k_inc_counter, dim3(1), dim3(64), 0, s, h->rg_t_dev);
    h->rg_capturing = false;
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(s, &g);
    if (rc || e != hipSuccess || !g) {
      for (int j = 0; j < k; j++) { (void)hipGraphExecDestroy(h->rg_exec[j]); (void)hipGraphDestroy(h->rg_graph[j]); }
      if (g) (void)hipGraphDestroy(g);
      return rc ? rc : fail(D3IL_EHIP, std::string("graph_rollout: stream capture failed: ") + hipGetErrorString(e));
    }
    h->rg_graph[k] = g;
    e = hipGraphInstantiate(&h->rg_exec[k], g, nullptr, nullptr, 0);
    if (e != hipSuccess) {
      for (int j = 0; j < k; j++) { (void)hipGraphExecDestroy(h->rg_exec[j]); (void)hipGraphDestroy(h->rg_graph[j]); }
      (void)hipGraphDestroy(g);
      return fail(D3IL_EHIP, std::string("graph_rollout: hipGraphInstantiate: ") + hipGetErrorString(e));
    }
  }
  h->rg_seed = seed; h->rg_off = env_offset; h->rg_actions = actions; h->rg_counts = counts; h->rg_stream = s; h->rg_next_t = t;
  h->rg_launched = h->rg_drained = 0; h->rg_ready = true;
  return D3IL_OK;
}
// captures the graphs of the NEXT d3il_random_rollout_step calls with these arguments without launching anything (a harness does this outside its timed region);
// a no-op without option graph_rollout
int d3il_random_rollout_prepare(d3il_handle h, uint64_t seed, uint64_t env_offset, uint32_t t, double* actions, int64_t* episode_counts_device, void* stream) {
  if (!h || !actions) return fail(D3IL_EINVAL, "d3il_random_rollout_prepare: null argument");
  if (!(h->rg_enabled && h->task_id == D3IL_TASK_AVOIDING && stream != nullptr && h->started)) return D3IL_OK;
  hipStream_t s = (hipStream_t)stream;
  HIPCHK(hipSetDevice(h->device));
  if (h->rg_ready && (h->rg_seed != seed || h->rg_off != env_offset || h->rg_actions != actions || h->rg_counts != episode_counts_device || h->rg_stream != s ||
                      h->rg_timing != h->timing)) {
    HIPCHK(hipStreamSynchronize(h->rg_stream));
    rg_drop(h);
  }
  if (!h->rg_ready) return rg_capture(h, seed, env_offset, t, actions, episode_counts_device, s);
  return D3IL_OK;
}
int d3il_random_rollout_step(d3il_handle h, uint64_t seed, uint64_t env_offset, uint32_t t, double* actions, int64_t* episode_counts_device, void* stream) {
  if (h && h->rg_enabled && h->task_id == D3IL_TASK_AVOIDING && stream != nullptr && actions && h->started) {
    hipStream_t s = (hipStream_t)stream;
    if (int rc_ = drop_prepared_action(h)) return rc_;      // (the graph path draws its own action: a pending fused-tail draw ends here)
    if (int rc = d3il_random_rollout_prepare(h, seed, env_offset, t, actions, episode_counts_device, stream)) return rc;
    if (t != h->rg_next_t) { HIPCHK(hipMemcpyAsync(h->rg_t_dev, &t, sizeof(unsigned), hipMemcpyHostToDevice, s)); HIPCHK(hipStreamSynchronize(s)); }      // the caller jumped in time
    // Launch durations: event-record nodes inside a captured graph give no usable timestamps with this runtime (hipEventElapsedTime: invalid resource
    // handle), so with timing on every RG_SAMPLE-th step goes through the uncaptured sequence with the event ring around its step launch - a uniform
    // 1-in-RG_SAMPLE sample of the launches of the timed region (d3il_timing_stats reports how many).
    if (h->timing && h->rg_launched % RG_SAMPLE == RG_SAMPLE - 1) {
      if (int rc = d3il_policy_action(h, seed, env_offset, t, actions, stream)) return rc;
      if (int rc = d3il_step(h, actions, stream)) return rc;
      if (int rc = d3il_auto_reset(h, episode_counts_device, stream)) return rc;
      hipLaunchKernelGGL(k_inc_counter, dim3(1), dim3(64), 0, s, h->rg_t_dev);
      HIPCHK(hipGetLastError());
    } else HIPCHK(hipGraphLaunch(h->rg_exec[h->rg_launched % RG_SLOTS], s));
    h->rg_launched++; h->rg_next_t = t + 1;
    return D3IL_OK;
  }
  if (h && h->fuse_tail && h->task_id == D3IL_TASK_AVOIDING && actions && episode_counts_device && h->started) {
    // two launches per step: the step kernel and k_avoiding_tail (mask, tally, auto-reset, the NEXT step's action); the first call of a sequence (or one
    // whose arguments / step counter do not continue the previous call) starts with the separate policy kernel
    const bool cont = h->prep_valid && h->prep_t == t && h->prep_seed == seed && h->prep_off == env_offset && h->prep_actions == actions && h->prep_stream == (hipStream_t)stream;
    if (!h->d_des_before) { HIPCHK(hipSetDevice(h->device)); HIPCHK(hipMalloc(&h->d_des_before, (size_t)h->stride * 2 * sizeof(double))); }
    if (cont) h->prep_valid = 0;          // consumed: the calls below must not put the pose back
    else { if (int rc = d3il_policy_action(h, seed, env_offset, t, actions, stream)) return rc; }      // (d3il_policy_action drops a stale prepared action first)
    if (int rc = d3il_step(h, actions, stream)) return rc;
    d3il_buffers& b = h->buf;
    hipLaunchKernelGGL(k_avoiding_tail, dim3(h->stride / WAVE), dim3(WAVE), 0, (hipStream_t)stream, h->d_init_qpos, b.state, b.flags, b.step_count, b.obs, b.done, b.success, b.mode,
                       b.policy_des, (long long*)episode_counts_device, h->d_mask, h->tally_ctx, (long long*)h->tally_table, h->tally_nctx, actions, h->d_des_before,
                       (unsigned long long)seed, (unsigned long long)env_offset, t + 1u, h->n, h->stride);
    HIPCHK(hipGetLastError());
    h->prep_valid = true; h->prep_t = t + 1u; h->prep_seed = seed; h->prep_off = env_offset; h->prep_actions = actions; h->prep_stream = (hipStream_t)stream;
    return D3IL_OK;
  }
  if (int rc = d3il_policy_action(h, seed, env_offset, t, actions, stream)) return rc;
  if (int rc = d3il_step(h, actions, stream)) return rc;
  return d3il_auto_reset(h, episode_counts_device, stream);
}
int d3il_last_step_ms(d3il_handle h, float* ms) {
  if (!h || !ms) return fail(D3IL_EINVAL, "d3il_last_step_ms: null argument");
  if (!h->ev_valid) return fail(D3IL_ESTATE, "d3il_last_step_ms: timing not enabled or no step recorded");
  HIPCHK(hipEventSynchronize(h->ev1));
  HIPCHK(hipEventElapsedTime(ms, h->ev0, h->ev1));
  return D3IL_OK;
}
int d3il_debug_wave_stats(uint64_t* out, int nwaves, int reset) {
#if defined(D3IL_DEVICE_STATS)
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(d3il::g_dev_wave), (size_t)nwaves * 10 * sizeof(unsigned long long)));
  if (reset) { static unsigned long long z[4096][10]; HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(d3il::g_dev_wave), z, sizeof z)); }
  return D3IL_OK;
#else
  (void)out; (void)nwaves; (void)reset;
  return fail(D3IL_EUNSUPPORTED, "d3il_debug_wave_stats: library built without D3IL_DEVICE_STATS");
#endif
}

/* diagnostics build only: per-workgroup event counters of the generic engine's solver (PUSH_CNT), [nwaves][8] */
int d3il_debug_wave_counts(uint64_t* out, int nwaves, int reset) {
#if defined(D3IL_DEVICE_STATS)
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(d3il::g_dev_cnt), (size_t)nwaves * 8 * sizeof(unsigned long long)));
  if (reset) { static unsigned long long z[4096][8]; HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(d3il::g_dev_cnt), z, sizeof z)); }
  return D3IL_OK;
#else
  (void)out; (void)nwaves; (void)reset;
  return fail(D3IL_EUNSUPPORTED, "d3il_debug_wave_counts: library built without D3IL_DEVICE_STATS");
#endif
}

/* diagnostics build only: copies (and optionally clears) the device path counters; returns EUNSUPPORTED otherwise */
int d3il_debug_stats(uint64_t* out32, int reset) {
#if defined(D3IL_DEVICE_STATS)
  unsigned long long tmp[32];
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpyFromSymbol(tmp, HIP_SYMBOL(d3il::g_dev_stats), sizeof tmp));
  for (int i = 0; i < 32; i++) out32[i] = tmp[i];
  if (reset) { unsigned long long z[32] = {0}; HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(d3il::g_dev_stats), z, sizeof z)); }
  return D3IL_OK;
#else
  (void)out32; (void)reset;
  return fail(D3IL_EUNSUPPORTED, "d3il_debug_stats: library built without D3IL_DEVICE_STATS");
#endif
}

/* diagnostics: one environment's column of the solver scratch area (contact records of its last sub-step; layout in the task's
 * *_step.h) copied to the host */
int d3il_debug_scratch(d3il_handle h, int env, double* out, int count) {
  if (!h || !out) return fail(D3IL_EINVAL, "d3il_debug_scratch: null argument");
  if (!h->d_scratch || env < 0 || env >= h->n) return fail(D3IL_EINVAL, "d3il_debug_scratch: no scratch area / env out of range");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipDeviceSynchronize());
  if (h->task_id == D3IL_TASK_STACKING || h->task_id == D3IL_TASK_ALIGNING) {     // contiguous per environment (the cooperative engine)
    if (count > SG_SIZE) return fail(D3IL_EINVAL, "d3il_debug_scratch: count exceeds the environment's scratch area");
    HIPCHK(hipMemcpy(out, h->d_scratch + (size_t)env * SG_SIZE, (size_t)count * sizeof(double), hipMemcpyDeviceToHost));
    return D3IL_OK;
  }
  if (gen_task(h->task_id)) {     // blocked by workgroup: [environment / GEN_LANES][field][environment % GEN_LANES] (gen_step.h GRS)
    if (count > GG_SIZE) return fail(D3IL_EINVAL, "d3il_debug_scratch: count exceeds the environment's scratch area");
    // [workgroup][field pair][column][2]: the even and the odd fields are two strided copies
    const double* base = h->d_scratch + (size_t)(env / GEN_LANES) * GG_BLOCK * GEN_LANES + 2 * (env % GEN_LANES);
    HIPCHK(hipMemcpy2D(out, 2 * sizeof(double), base, (size_t)2 * GEN_LANES * sizeof(double), sizeof(double), (size_t)(count + 1) / 2, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy2D(out + 1, 2 * sizeof(double), base + 1, (size_t)2 * GEN_LANES * sizeof(double), sizeof(double), (size_t)count / 2, hipMemcpyDeviceToHost));
    return D3IL_OK;
  }
  HIPCHK(hipMemcpy2D(out, sizeof(double), h->d_scratch + env, (size_t)h->stride * sizeof(double), sizeof(double), (size_t)count, hipMemcpyDeviceToHost));
  return D3IL_OK;
}

int d3il_set_option(d3il_handle h, const char* name, int value) {
  if (!h || !name) return fail(D3IL_EINVAL, "d3il_set_option: null argument");
  if (h->rg_ready) { HIPCHK(hipDeviceSynchronize()); rg_drop(h); }      // an option may change what a step launches
  if (std::strcmp(name, "graph_rollout") == 0) { if (int rc_ = drop_prepared_action(h)) return rc_; h->rg_enabled = value != 0; return D3IL_OK; }
  if (std::strcmp(name, "fuse_rollout_tail") == 0) { if (int rc_ = drop_prepared_action(h)) return rc_; h->fuse_tail = value != 0; return D3IL_OK; }
  if (std::strcmp(name, "ik_fast_path") == 0) { h->fast = value != 0; return D3IL_OK; }
  if (std::strcmp(name, "solver_strict") == 0) { h->tol_mode = value != 0; return D3IL_OK; }
  if (std::strcmp(name, "stack_reset_coop") == 0) { h->stack_reset_coop = value != 0; return D3IL_OK; }
  if (std::strcmp(name, "split_waves") == 0) { h->split = value; return D3IL_OK; }
  if (std::strcmp(name, "serve_wave_max_workgroups") == 0) { if (value < 0) return fail(D3IL_EINVAL, "serve_wave_max_workgroups must be >= 0"); h->serve_max_wg = h->serve_avail ? value : 0; return D3IL_OK; }
  if (std::strcmp(name, "lds_pad_bytes") == 0) { h->lds_pad = value; return D3IL_OK; }
  if (std::strcmp(name, "lanes_per_wave") == 0) { if (value < 1 || value > WAVE) return fail(D3IL_EINVAL, "lanes_per_wave must be in 1..64"); h->lanes = value; return D3IL_OK; }
  return fail(D3IL_EINVAL, std::string("d3il_set_option: unknown option ") + name);
}

}  // extern "C"
