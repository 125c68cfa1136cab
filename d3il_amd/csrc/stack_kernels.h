// stack_kernels.h - HIP kernels of the Stacking task (included by rollout.hip).
//
// First, correctness-first shape: ONE LANE PER ENVIRONMENT, SK_LANES = 24 environments per workgroup (one wave, the other lanes
// idle): a lane's vectors, kinematic tables and the packed 27 x 27 Newton Hessian live in LDS (717 doubles per environment,
// lane-strided => conflict-free, 134.4 KiB per workgroup = one workgroup per CU), contact records in an HBM scratch area
// (36 doubles x 48 per environment, lane-strided => coalesced).  The arm is the gripper robot of panda_invisible.xml, its
// constants baked at build time (csrc/gen/stacking_consts.inc).  No controller wave: the task's control law is a joint PD.
#pragma once
#include "stack_step.h"

namespace d3il {

constexpr int STACK_LDS = ST_SIZE * SK_LANES * 8;

__device__ __forceinline__ void stack_load(const double* __restrict__ state, const unsigned* __restrict__ flags, const int* __restrict__ steps, int stride, int e, StackState& ss) {
  const double* s = state + e;
  EnvState& st = ss.arm;
  for (int i = 0; i < NDOF; i++) st.q[i] = s[(size_t)i * stride];
  for (int i = 0; i < NDOF; i++) st.v[i] = s[(size_t)(9 + i) * stride];
  for (int i = 0; i < NARM; i++) st.bias[i] = s[(size_t)(18 + i) * stride];
  for (int i = 0; i < 3; i++) st.tcp[i] = s[(size_t)(25 + i) * stride];
  int k = SK_STATE_BOX;
  for (int b = 0; b < SK_NB; b++) {
    for (int i = 0; i < 3; i++) ss.box[b].pos[i] = s[(size_t)(k++) * stride];
    for (int i = 0; i < 4; i++) ss.box[b].quat[i] = s[(size_t)(k++) * stride];
    for (int i = 0; i < 6; i++) ss.box[b].vel[i] = s[(size_t)(k++) * stride];
  }
  for (int i = 0; i < SK_NV; i++) ss.warm[i] = s[(size_t)(SK_STATE_WARM + i) * stride];
  st.flags = flags[e]; st.step = steps[e];
}
__device__ __forceinline__ void stack_store(double* __restrict__ state, unsigned* __restrict__ flags, int* __restrict__ steps, int stride, int e, const StackState& ss) {
  double* s = state + e;
  const EnvState& st = ss.arm;
  for (int i = 0; i < NDOF; i++) s[(size_t)i * stride] = st.q[i];
  for (int i = 0; i < NDOF; i++) s[(size_t)(9 + i) * stride] = st.v[i];
  for (int i = 0; i < NARM; i++) s[(size_t)(18 + i) * stride] = st.bias[i];
  for (int i = 0; i < 3; i++) s[(size_t)(25 + i) * stride] = st.tcp[i];
  int k = SK_STATE_BOX;
  for (int b = 0; b < SK_NB; b++) {
    for (int i = 0; i < 3; i++) s[(size_t)(k++) * stride] = ss.box[b].pos[i];
    for (int i = 0; i < 4; i++) s[(size_t)(k++) * stride] = ss.box[b].quat[i];
    for (int i = 0; i < 6; i++) s[(size_t)(k++) * stride] = ss.box[b].vel[i];
  }
  for (int i = 0; i < SK_NV; i++) s[(size_t)(SK_STATE_WARM + i) * stride] = ss.warm[i];
  flags[e] = st.flags; steps[e] = st.step;
}

// env.step(action[8]) for the Stacking task (stacking.py:331-393): 7 joint targets + gripper command
__global__ __launch_bounds__(WAVE) void k_stacking_step(double* __restrict__ state, unsigned* __restrict__ flags, int* __restrict__ steps,
                                                        const double* __restrict__ actions, float* __restrict__ obs, unsigned char* __restrict__ done,
                                                        unsigned char* __restrict__ success, unsigned short* __restrict__ mode, double* __restrict__ info,
                                                        double* __restrict__ scratch, int n, int stride, int n_substeps, int max_steps) {
  extern __shared__ double smem[];
  const int lane = threadIdx.x;
  const int e = blockIdx.x * SK_LANES + lane;
  if (lane >= SK_LANES || e >= n) return;
  StackScratch sc{(sk_lds_double*)(smem + lane), (sk_glb_double*)(scratch + e), stride};
  StackState ss;
  stack_load(state, flags, steps, stride, e, ss);
  double act[SK_ACT];
  bool bad = false;
#pragma unroll
  for (int k = 0; k < SK_ACT; k++) { act[k] = actions[(size_t)e * SK_ACT + k]; unsigned long long b; __builtin_memcpy(&b, &act[k], 8); bad = bad || ((b >> 52) & 0x7ffull) == 0x7ffull; }
  if (bad) {    // NaN / Inf action: hold the current joints with an open gripper; the lane is flagged and terminated
#pragma unroll
    for (int k = 0; k < NARM; k++) act[k] = ss.arm.q[k];
    act[7] = 1.0;
  }
  float o[SK_OBS]; unsigned char dn = 0; double md = 0;
  stack_env_step(kStackingConsts, g_stack_consts, ss, sc, act, o, &dn, &md, n_substeps, max_steps);
  if (bad) ss.arm.flags |= F_SOLVER_FAIL | F_TERMINATED;
  stack_store(state, flags, steps, stride, e, ss);
#pragma unroll
  for (int k = 0; k < SK_OBS; k++) obs[(size_t)SK_OBS * e + k] = o[k];
  done[e] = dn; success[e] = (ss.arm.flags & F_SUCCESS) ? 1 : 0; mode[e] = (unsigned short)stack_mode_code(ss.arm.flags);
  info[e] = md;
}

// env.reset(random=False, context) for masked environments; contexts: f64 [n][21] = 3 x (pos3, quat4), red green blue
__global__ __launch_bounds__(WAVE) void k_stacking_reset(const double* __restrict__ init_qpos, const unsigned char* __restrict__ mask, const double* __restrict__ contexts,
                                                         double* __restrict__ state, unsigned* __restrict__ flags, int* __restrict__ steps, float* __restrict__ obs,
                                                         unsigned char* __restrict__ done, unsigned char* __restrict__ success, unsigned short* __restrict__ mode,
                                                         double* __restrict__ info, double* __restrict__ scratch, int n, int stride) {
  extern __shared__ double smem[];
  const int lane = threadIdx.x;
  const int e = blockIdx.x * SK_LANES + lane;
  if (lane >= SK_LANES || e >= n) return;
  if (mask && !mask[e]) return;
  StackScratch sc{(sk_lds_double*)(smem + lane), (sk_glb_double*)(scratch + e), stride};
  StackState ss;
  double iq[NARM], ctx[21];
#pragma unroll
  for (int k = 0; k < NARM; k++) iq[k] = init_qpos[k];
  for (int k = 0; k < 21; k++) ctx[k] = contexts[(size_t)e * 21 + k];
  float o[SK_OBS];
  stack_env_reset(kStackingConsts, g_stack_consts, ss, sc, iq, ctx, o);
  stack_store(state, flags, steps, stride, e, ss);
#pragma unroll
  for (int k = 0; k < SK_OBS; k++) obs[(size_t)SK_OBS * e + k] = o[k];
  done[e] = 0; success[e] = 0; mode[e] = 0; info[e] = 0;
}

}  // namespace d3il
