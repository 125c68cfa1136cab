// stack_kernels.h - HIP kernels of the Stacking task (included by rollout.hip).
//
// A workgroup is ONE wave that owns SK_LANES environments (4: 1024 workgroups for 4096 environments, one per SIMD).  The per-lane
// parts of a sub-step (arm dynamics, collision, limit rows, integration: one lane per environment, the other lanes idle) alternate
// with the wave-cooperative constraint solve (sk_solve_coop: all 64 lanes on one environment, the workgroup's environments one after
// the other, so an environment iterates exactly as long as IT needs and the lanes of a wave never wait for each other's Newton
// iterations).  LDS: 717 doubles per environment (vectors, kinematic tables, packed 27 x 27 Hessian; contiguous per environment) + the
// wave's contact-row area (60 x 32 doubles) = 38.3 KiB per workgroup, four workgroups per CU; contact records in an HBM scratch
// area (36 doubles x 32 per environment, contiguous per environment).  The arm is the gripper robot of panda_invisible.xml, its
// constants baked at build time (csrc/gen/stacking_consts.inc).  No controller wave: the task's control law is a joint PD.
// The reset kernel runs the one-lane solver (sk_solve) - it is the host build's code path and off the hot path.
#pragma once
#include "stack_step.h"

namespace d3il {

constexpr int STACK_LDS_RESET = ST_SIZE * SK_LANES * 8;
constexpr int STACK_LDS = (ST_SIZE * SK_LANES + SKC_JSIZE) * 8;

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
  const bool live = lane < SK_LANES && e < n;
  const int le = live ? lane : 0;
  const size_t ee = live ? e : 0;
  const StackScratch sc{(sk_lds_double*)(smem + le * ST_SIZE), (sk_glb_double*)(scratch + ee * SG_SIZE)};
  sk_lds_double* Jw = (sk_lds_double*)(smem + SK_LANES * ST_SIZE);
  StackState ss;
  double act[SK_ACT];
  bool bad = false, open = true;
  float o[SK_OBS]; unsigned char dn = 0; double md = 0;
  if (live) {
    stack_load(state, flags, steps, stride, e, ss);
#pragma unroll
    for (int k = 0; k < SK_ACT; k++) act[k] = actions[(size_t)e * SK_ACT + k];
    bad = action_is_bad(actions + (size_t)e * SK_ACT, SK_ACT);      // integer test on the words in memory (see panda_step.h)
    if (bad) {    // NaN / Inf action: hold the current joints with an open gripper; the lane is flagged and terminated
#pragma unroll
      for (int k = 0; k < NARM; k++) act[k] = ss.arm.q[k];
      act[7] = 1.0;
    }
    for (int i = 0; i < SK_NV; i++) SL(ST_X + i) = ss.warm[i];     // the warm start lives in the t area between the sub-steps
    open = stack_env_begin(g_stack_consts, ss, act, o, &dn, max_steps);
  }
#pragma clang loop unroll(disable)
  for (int s = 0; s < n_substeps; s++) {
    int ncon = 0; bool any_lim = false, over = false;
    unsigned has = 0;
    if (live) {
      double tau[NARM], ff[NFING];
      stack_control(kStackingConsts, ss.arm, act, open ? 0.04 : 0.0, !open, tau, ff);
      stack_pre_kin(kStackingConsts, g_stack_consts, ss, sc, tau, ff);
    }
    __syncthreads();
    SK_TIC;
    sk_collide_coop(g_stack_consts, (sk_lds_double*)smem, Jw, (sk_glb_double*)(scratch + (size_t)blockIdx.x * SK_LANES * SG_SIZE), lane, live ? 1 : 0, ncon, has, over);
    if (live) {
      SK_TOC(1);
      if (over) ss.arm.flags |= SKF_CON_OVERFLOW;
      stack_pre_finish<true>(kStackingConsts, g_stack_consts, ss, sc, ncon, has, any_lim);
    }
    const int need = (live && (ncon > 0 || any_lim)) ? 1 : 0;
    const int warm = (live && (ss.arm.flags & SKF_WARM_VALID)) ? 1 : 0;
    __syncthreads();
#pragma clang loop unroll(disable)
    for (int e2 = 0; e2 < SK_LANES; e2++) {
      if (!__builtin_amdgcn_readlane(need, e2)) continue;
      const bool ok = sk_solve_coop(g_stack_consts, (sk_lds_double*)(smem + e2 * ST_SIZE), Jw,
                                    (sk_glb_double*)(scratch + ((size_t)blockIdx.x * SK_LANES + e2) * SG_SIZE), lane,
                                    __builtin_amdgcn_readlane(ncon, e2), __builtin_amdgcn_readlane(warm, e2) != 0);
      if (!ok && lane == e2) ss.arm.flags |= F_SOLVER_FAIL;
    }
    __syncthreads();
    if (live) stack_substep_post<true>(kStackingConsts, g_stack_consts, ss, sc);
  }
  if (!live) return;
  for (int i = 0; i < SK_NV; i++) ss.warm[i] = SL(ST_X + i);
  stack_env_end(g_stack_consts, ss, &md);
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
  const StackScratch sc{(sk_lds_double*)(smem + lane * ST_SIZE), (sk_glb_double*)(scratch + (size_t)e * SG_SIZE)};
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
