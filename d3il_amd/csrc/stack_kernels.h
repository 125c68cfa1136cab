// stack_kernels.h - HIP kernels of the Stacking task (included by rollout.hip).
//
// A workgroup is ONE wave that owns SK_LANES environments (4: 1024 workgroups for 4096 environments, one per SIMD).  The per-lane
// parts of a sub-step (arm dynamics, collision, limit rows, integration: one lane per environment, the other lanes idle) alternate
// with the wave-cooperative constraint solve (sk_solve_dual: the 64 lanes work on TWO environments at a time, one per half wave - a
// 27-dof system with at most 32 contacts fits 32 lanes -, the workgroup's two pairs one after the other; a half that has converged
// idles until the other one has).  LDS (layout: stack_step.h SE_*): the workgroup's shared Hessian + wrench-table / staging area, and per environment its
// vectors, kinematic tables, STATE and compact contact records = 38.2 KiB per workgroup, four workgroups per CU.  An environment's
// state is in registers only inside a phase; between the phases of a sub-step everything is LDS resident and nothing goes through HBM
// (the round-2 kernel kept 134 VGPRs of state live across the cooperative phases and its contact records in an HBM scratch area).  The arm is the gripper robot of panda_invisible.xml, its
// constants baked at build time (csrc/gen/stacking_consts.inc).  No controller wave: the task's control law is a joint PD.
// The reset kernel runs the one-lane solver (sk_solve) - it is the host build's code path and off the hot path.
#pragma once
#include "stack_step.h"
#include "align_step.h"

namespace d3il {

constexpr int STACK_LDS_RESET = ST_SIZE * SK_LANES * 8;
constexpr int STACK_LDS = (SKC_SHARED + SE_SIZE * SK_LANES) * 8;
static_assert(STACK_LDS <= 40 * 1024, "four workgroups per CU (one per SIMD) need <= 40 KiB of LDS each");

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

// ---- the environment's state between the phases of a sub-step: LDS (velocities in ST_VEL, box positions in ST_BP, the rest in SE_*)
__device__ __forceinline__ void sk_state_to_lds(sk_lds_double* t, const StackState& ss) {
  for (int k = 0; k < NDOF; k++) { t[SE_Q + k] = ss.arm.q[k]; t[ST_VEL + SK_ARM0 + k] = ss.arm.v[k]; }
  for (int k = 0; k < NARM; k++) t[SE_BIAS + k] = ss.arm.bias[k];
  for (int k = 0; k < 3; k++) t[SE_TCP + k] = ss.arm.tcp[k];
  for (int b = 0; b < SK_NB; b++) {
    for (int k = 0; k < 3; k++) t[ST_BP + 3 * b + k] = ss.box[b].pos[k];
    for (int k = 0; k < 4; k++) t[SE_BQ + 4 * b + k] = ss.box[b].quat[k];
    for (int k = 0; k < 6; k++) t[ST_VEL + 6 * b + k] = ss.box[b].vel[k];
  }
}
__device__ __forceinline__ void sk_state_from_lds(const sk_lds_double* t, StackState& ss) {
  for (int k = 0; k < NDOF; k++) { ss.arm.q[k] = t[SE_Q + k]; ss.arm.v[k] = t[ST_VEL + SK_ARM0 + k]; }
  for (int k = 0; k < NARM; k++) ss.arm.bias[k] = t[SE_BIAS + k];
  for (int k = 0; k < 3; k++) ss.arm.tcp[k] = t[SE_TCP + k];
  for (int b = 0; b < SK_NB; b++) {
    for (int k = 0; k < 3; k++) ss.box[b].pos[k] = t[ST_BP + 3 * b + k];
    for (int k = 0; k < 4; k++) ss.box[b].quat[k] = t[SE_BQ + 4 * b + k];
    for (int k = 0; k < 6; k++) ss.box[b].vel[k] = t[ST_VEL + 6 * b + k];
  }
}
// phase 1 (lane = environment): control law + arm forward pass, kinematic tables, smooth accelerations
template <int V>
__device__ __forceinline__ void sk_phase_pre(sk_lds_double* t, sk_glb_double* g, unsigned& flags, bool open, const sk_lds_double* trig_lds) {
  double tau[NARM], ff[NFING];
  const StackScratch sc{t, g};
  if constexpr (V != SKV_STACKING) {
    // CartPosQuatImpedenceController (IKControllers.py:163-323): the virtual joint target advances open loop by three damped least-squares
    // iterations per sub-step, then the joint PD law with gravity compensation; fingers commanded open (gym_env_wrapper.py:67).
    // The controller runs FIRST, with nothing of the environment's state in registers yet (it needs the joint angles only when the target is
    // initialised): its three IK iterations are the longest dependent chain of the sub-step, and with the 60 doubles of the state live across
    // them the chain ran through scratch reloads (round 6: the per-environment phases were two thirds of the Aligning sub-step)
    double ikq[NARM], ikqd[NARM];
    const bool hold = V == SKV_ALIGNING && t[ST_TIPR + SV_HOLD] != 0.0;      // the sub-step of env.reset(): joint PD hold at init_qpos (in the ik_q slots), fingers at 1 mm
    SK_TIC;
    {
      double des[7], vwarm[7], cq[NARM];
      for (int k = 0; k < NARM; k++) { ikq[k] = t[ST_TIPR + SV_IKQ + k]; ikqd[k] = t[ST_TIPR + SV_IKQD + k]; des[k] = t[ST_TIPR + SV_DES + k]; vwarm[k] = t[ST_TIPR + SV_VWARM + k]; cq[k] = t[SE_Q + k]; }
      if (!hold) {
        ik_update<true>(kAvoidingConsts, des, des + 3, cq, flags, ikq, ikqd, vwarm);
        for (int k = 0; k < NARM; k++) { t[ST_TIPR + SV_IKQ + k] = ikq[k]; t[ST_TIPR + SV_IKQD + k] = ikqd[k]; t[ST_TIPR + SV_VWARM + k] = vwarm[k]; }
      }
    }
    SK_TOC(1);      // (diagnostics build: the controller's share of the per-environment phase)
    asm volatile("" ::: "memory");      // the state is loaded AFTER the controller, not hoisted above it
    StackState ss;
    sk_state_from_lds(t, ss);
    ss.arm.flags = flags;
    double trig[2 * NARM];
    for (int k = 0; k < 2 * NARM; k++) trig[k] = trig_lds[k];
    push_control(kAvoidingConsts, ss.arm, ikq, ikqd, hold ? 0.001 : 0.04, false, tau, ff);
    stack_pre_kin<V>(kAvoidingConsts, g_stack_consts, ss, sc, tau, ff, trig);
    for (int k = 0; k < NARM; k++) t[SE_BIAS + k] = ss.arm.bias[k];
    for (int k = 0; k < 3; k++) t[SE_TCP + k] = ss.arm.tcp[k];
    flags = ss.arm.flags;
  } else {
    StackState ss;
    sk_state_from_lds(t, ss);
    double trig[2 * NARM];
    for (int k = 0; k < 2 * NARM; k++) trig[k] = trig_lds[k];
    ss.arm.flags = flags;
    double act[NARM];
    for (int k = 0; k < NARM; k++) act[k] = t[SE_ACT + k];
    stack_control(kStackingConsts, ss.arm, act, open ? 0.04 : 0.0, !open, tau, ff);
    stack_pre_kin<SKV_STACKING>(kStackingConsts, g_stack_consts, ss, sc, tau, ff, trig);
    for (int k = 0; k < NARM; k++) t[SE_BIAS + k] = ss.arm.bias[k];
    for (int k = 0; k < 3; k++) t[SE_TCP + k] = ss.arm.tcp[k];
    flags = ss.arm.flags;
  }
}
// phase 3 (lane = environment): joint-limit rows, start point of the solver, "does the solver run"
template <int V>
__device__ __forceinline__ void sk_phase_mid(sk_lds_double* t, sk_glb_double* g, unsigned& flags) {
  StackState ss;
  for (int k = 0; k < NDOF; k++) { ss.arm.q[k] = t[SE_Q + k]; ss.arm.v[k] = t[ST_VEL + SK_ARM0 + k]; }
  ss.arm.flags = flags;
  const int ncon = (int)t[SE_NCON];
  if (t[SE_NEED] != 0.0) flags |= SKF_CON_OVERFLOW;      // the collision phase dropped contacts
  unsigned has = 0;
  for (int ci = 0; ci < ncon; ci++) {
    const int meta = (int)t[SE_REC + ci * SREC2 + 7];
    const int ba = sk_blk_of(meta & 15);
    has |= 1u << sk_blk_of((meta >> 4) & 15); if (ba >= 0) has |= 1u << ba;
  }
  bool any_lim = false;
  const StackScratch sc{t, g};
  if constexpr (V != SKV_STACKING) stack_pre_finish<true>(kAvoidingConsts, g_stack_consts, ss, sc, ncon, has, any_lim);
  else stack_pre_finish<true>(kStackingConsts, g_stack_consts, ss, sc, ncon, has, any_lim);
  t[SE_NEED] = (ncon > 0 || any_lim) ? 1.0 : 0.0;
}
// phase 5 (lane = environment): mj_Euler
template <int V>
__device__ __forceinline__ void sk_phase_post(sk_lds_double* t, sk_glb_double* g, unsigned& flags, sk_lds_double* trig_lds) {
  StackState ss;
  sk_state_from_lds(t, ss);
  ss.arm.flags = flags;
  const StackScratch sc{t, g};
  double q_old[NARM];
  for (int k = 0; k < NARM; k++) q_old[k] = ss.arm.q[k];
  if constexpr (V != SKV_STACKING) stack_substep_post<true>(kAvoidingConsts, g_stack_consts, ss, sc);
  else stack_substep_post<true>(kStackingConsts, g_stack_consts, ss, sc);
  for (int k = 0; k < NARM; k++) {      // |dq| = h |v| <= 1e-3 x a few rad/s: the increment that was actually applied
    double sn = trig_lds[k], cs = trig_lds[NARM + k];
    trig_advance(ss.arm.q[k] - q_old[k], sn, cs);
    trig_lds[k] = sn; trig_lds[NARM + k] = cs;
  }
  for (int k = 0; k < NDOF; k++) { t[SE_Q + k] = ss.arm.q[k]; t[ST_VEL + SK_ARM0 + k] = ss.arm.v[k]; }
  for (int b = 0; b < SK_NB; b++) {
    for (int k = 0; k < 3; k++) t[ST_BP + 3 * b + k] = ss.box[b].pos[k];
    for (int k = 0; k < 4; k++) t[SE_BQ + 4 * b + k] = ss.box[b].quat[k];
    for (int k = 0; k < 6; k++) t[ST_VEL + 6 * b + k] = ss.box[b].vel[k];
  }
  flags = ss.arm.flags;
}

__device__ __forceinline__ void align_load(const double* __restrict__ state, const unsigned* __restrict__ flags, const int* __restrict__ steps, int stride, int e, AlignState& as) {
  const double* s = state + e;
  EnvState& st = as.arm;
  for (int i = 0; i < NDOF; i++) st.q[i] = s[(D3IL_STATE_QPOS + i) * (size_t)stride];
  for (int i = 0; i < NDOF; i++) st.v[i] = s[(D3IL_STATE_QVEL + i) * (size_t)stride];
  for (int i = 0; i < NARM; i++) st.bias[i] = s[(D3IL_STATE_BIAS + i) * (size_t)stride];
  for (int i = 0; i < 3; i++) st.tcp[i] = s[(D3IL_STATE_TCP + i) * (size_t)stride];
  for (int i = 0; i < NARM; i++) st.ikq[i] = s[(D3IL_STATE_IK_Q + i) * (size_t)stride];
  for (int i = 0; i < NARM; i++) st.ikqd[i] = s[(D3IL_STATE_IK_QD + i) * (size_t)stride];
  int k = AL_STATE_BOX;
  for (int i = 0; i < 3; i++) as.box.pos[i] = s[(size_t)(k++) * stride];
  for (int i = 0; i < 4; i++) as.box.quat[i] = s[(size_t)(k++) * stride];
  for (int i = 0; i < 6; i++) as.box.vel[i] = s[(size_t)(k++) * stride];
  for (int i = 0; i < 7; i++) as.target[i] = s[(size_t)(AL_STATE_TARGET + i) * stride];
  st.flags = flags[e]; st.step = steps[e];
}
__device__ __forceinline__ void align_store(double* __restrict__ state, unsigned* __restrict__ flags, int* __restrict__ steps, int stride, int e, const AlignState& as) {
  double* s = state + e;
  const EnvState& st = as.arm;
  for (int i = 0; i < NDOF; i++) s[(D3IL_STATE_QPOS + i) * (size_t)stride] = st.q[i];
  for (int i = 0; i < NDOF; i++) s[(D3IL_STATE_QVEL + i) * (size_t)stride] = st.v[i];
  for (int i = 0; i < NARM; i++) s[(D3IL_STATE_BIAS + i) * (size_t)stride] = st.bias[i];
  for (int i = 0; i < 3; i++) s[(D3IL_STATE_TCP + i) * (size_t)stride] = st.tcp[i];
  for (int i = 0; i < NARM; i++) s[(D3IL_STATE_IK_Q + i) * (size_t)stride] = st.ikq[i];
  for (int i = 0; i < NARM; i++) s[(D3IL_STATE_IK_QD + i) * (size_t)stride] = st.ikqd[i];
  int k = AL_STATE_BOX;
  for (int i = 0; i < 3; i++) s[(size_t)(k++) * stride] = as.box.pos[i];
  for (int i = 0; i < 4; i++) s[(size_t)(k++) * stride] = as.box.quat[i];
  for (int i = 0; i < 6; i++) s[(size_t)(k++) * stride] = as.box.vel[i];
  for (int i = 0; i < 7; i++) s[(size_t)(AL_STATE_TARGET + i) * stride] = as.target[i];
  flags[e] = st.flags; steps[e] = st.step;
}
// the engine's three blocks for the Aligning task: block 0 = the compound body, blocks 1 and 2 parked and inert
__device__ __forceinline__ void align_to_stack(const AlignState& as, StackState& ss) {
  ss.arm = as.arm; ss.box[0] = as.box;
  for (int b = 1; b < SK_NB; b++) {
    for (int k = 0; k < 3; k++) ss.box[b].pos[k] = 100.0 * b;
    ss.box[b].quat[0] = 1; ss.box[b].quat[1] = ss.box[b].quat[2] = ss.box[b].quat[3] = 0;
    for (int k = 0; k < 6; k++) ss.box[b].vel[k] = 0;
  }
}

// env.step(action[8]) for the Stacking task (stacking.py:331-393): 7 joint targets + gripper command
// The same kernel is env.reset() (stacking.py:449-481) when `reset` is set: the masked environments (reset_mask, NULL = all) are beamed to
// init_qpos with their boxes at `contexts` (f64 [n][21] = 3 x (pos3, quat4): red, green, blue), one physics sub-step runs under the joint PD
// hold with the fingers commanded open - through the same cooperative collision / solver phases as a step - and the observation of the new
// state is written.  Workgroups without a masked environment leave at once, so the auto-reset of the few episodes that end in a step costs
// one sub-step of one workgroup instead of a one-lane solve (0.8 ms per call in round 2).
template <int V>
__device__ __forceinline__ void coop_step_body(double* __restrict__ state, unsigned* __restrict__ flags, int* __restrict__ steps,
                                               const double* __restrict__ actions, float* __restrict__ obs, unsigned char* __restrict__ done,
                                               unsigned char* __restrict__ success, unsigned short* __restrict__ mode, double* __restrict__ info,
                                               double* __restrict__ scratch, int n, int stride, int n_substeps, int max_steps,
                                               const int reset, const unsigned char* __restrict__ reset_mask, const double* __restrict__ init_qpos,
                                               const double* __restrict__ contexts) {
  extern __shared__ double smem[];
  const int lane = threadIdx.x;
  const int e = blockIdx.x * SK_LANES + lane;
  const bool live = lane < SK_LANES && e < n && (!reset || !reset_mask || reset_mask[e] != 0);
  if (reset && !__any(live)) return;
  sk_lds_double* const sm = (sk_lds_double*)smem;
  sk_lds_double* const t = sk_env_view(sm, live ? lane : 0);
  sk_glb_double* const g = (sk_glb_double*)(scratch + (size_t)(live ? e : 0) * SG_SIZE);      // diagnostics words only (stats build)
  unsigned fl = 0; int step = 0;
  bool bad = false, open = true;
#if defined(D3IL_SK_POISON)
  // diagnostics build: every LDS word starts as a NaN, and at the start of every sub-step the areas that hold nothing live (the shared
  // solver / staging region, smooth accelerations, mass matrix, kinematic tables, limit rows, contact records) are poisoned again: a read of a
  // word this sub-step has not written shows up as a NaN in the state or as a solver failure instead of as a plausible stale number
  auto poison = [&](int q) { ((unsigned long long*)smem)[q] = 0x7ff8dead00000000ull; };
  for (int q = lane; q < STACK_LDS / 8; q += WAVE) poison(q);
  __syncthreads();
#endif
  if (V == SKV_STACKING && live && reset) {
    StackState ss;
    EnvState& st = ss.arm;
    for (int k = 0; k < NDOF; k++) { st.q[k] = k < NARM ? init_qpos[k] : 0.0; st.v[k] = 0; }
    {   // mj_forward at the beamed pose: qfrc_bias and TCP of that pass are what the first controller call reads
      DynOut dyn;
      dynamics(kStackingConsts, st.q, st.v, dyn);
      for (int k = 0; k < NARM; k++) st.bias[k] = dyn.bias[k];
      double tt[3]; mulE(dyn.R7, kStackingConsts.tcp7, tt);
      for (int k = 0; k < 3; k++) st.tcp[k] = dyn.p7[k] + tt[k];
    }
    for (int b = 0; b < SK_NB; b++) {
      for (int k = 0; k < 3; k++) ss.box[b].pos[k] = contexts[(size_t)e * 21 + 7 * b + k];
      for (int k = 0; k < 4; k++) ss.box[b].quat[k] = contexts[(size_t)e * 21 + 7 * b + 3 + k];
      for (int k = 0; k < 6; k++) ss.box[b].vel[k] = 0;
    }
    for (int i = 0; i < SK_NV; i++) t[ST_X + i] = 0;
    sk_state_to_lds(t, ss);
    for (int k = 0; k < NARM; k++) t[SE_ACT + k] = init_qpos[k];      // joint PD hold at init_qpos, open_fingers() (stacking.py:474)
    fl = 0; step = 0;
  } else if (live && V == SKV_ALIGNING && reset) {
    // Robot_Push_Env.reset(random=False, context) (aligning.py:344-372): robot beamed to init_qpos with closed fingers, the box at the context pose,
    // controllers replaced (ik_q / ik_qd cleared, IK_VALID off), ONE physics sub-step under the joint PD hold; contexts f64 [n][14] = box pos3
    // quat4 | target pos3 quat4
    AlignState as;
    EnvState& st = as.arm;
    for (int k = 0; k < NDOF; k++) { st.q[k] = k < NARM ? init_qpos[k] : 0.0; st.v[k] = 0; }
    for (int k = 0; k < NARM; k++) { st.ikq[k] = 0; st.ikqd[k] = 0; }
    {
      DynOut dyn;
      dynamics(kAvoidingConsts, st.q, st.v, dyn);
      for (int k = 0; k < NARM; k++) st.bias[k] = dyn.bias[k];
      double tt[3]; mulE(dyn.R7, kAvoidingConsts.tcp7, tt);
      for (int k = 0; k < 3; k++) st.tcp[k] = dyn.p7[k] + tt[k];
    }
    for (int k = 0; k < 3; k++) as.box.pos[k] = contexts[(size_t)e * AL_CTX + k];
    for (int k = 0; k < 4; k++) as.box.quat[k] = contexts[(size_t)e * AL_CTX + 3 + k];
    for (int k = 0; k < 6; k++) as.box.vel[k] = 0;
    for (int k = 0; k < 7; k++) as.target[k] = contexts[(size_t)e * AL_CTX + 7 + k];
    for (int i = 0; i < SK_NV; i++) t[ST_X + i] = 0;
    StackState ss;
    align_to_stack(as, ss);
    sk_state_to_lds(t, ss);
    for (int k = 0; k < NARM; k++) { t[ST_TIPR + SV_IKQ + k] = init_qpos[k]; t[ST_TIPR + SV_IKQD + k] = 0; t[ST_TIPR + SV_DES + k] = 0; t[ST_TIPR + SV_VWARM + k] = 0; }
    t[ST_TIPR + SV_HOLD] = 1.0;
    for (int k = 0; k < 7; k++) t[SE_ACT + k] = as.target[k];      // the target pose rides in the action slots of the environment region
    fl = 0; step = 0;
  } else if (live && V == SKV_ALIGNING) {
    // Robot_Push_Env.step (aligning.py:282-286 over gym_env_wrapper.py:45-100): set-point, observation / reward / done BEFORE the physics
    AlignState as;
    align_load(state, flags, steps, stride, e, as);
    double act[7], des[7];
#pragma unroll
    for (int k = 0; k < 7; k++) act[k] = actions[(size_t)e * 7 + k];
    bad = sanitize_action(act, actions + (size_t)e * 7);
    make_setpoint(act, des);
    float o[AL_OBS]; unsigned char dn = 0; double reward = 0;
    align_step_begin(g_align_task, as, o, &reward, &dn, max_steps);
#pragma unroll
    for (int k = 0; k < AL_OBS; k++) obs[(size_t)AL_OBS * e + k] = o[k];
    done[e] = dn; info[(size_t)stride + e] = reward;
    StackState ss;
    align_to_stack(as, ss);
    const double* sw = state + e + (size_t)AL_STATE_WARM * stride;      // warm start: box [6] arm [9]
    for (int i = 0; i < 6; i++) t[ST_X + i] = sw[(size_t)i * stride];
    for (int i = 6; i < 18; i++) t[ST_X + i] = 0;
    for (int i = 0; i < NDOF; i++) t[ST_X + SK_ARM0 + i] = sw[(size_t)(6 + i) * stride];
    sk_state_to_lds(t, ss);
    for (int k = 0; k < NARM; k++) { t[ST_TIPR + SV_IKQ + k] = as.arm.ikq[k]; t[ST_TIPR + SV_IKQD + k] = as.arm.ikqd[k]; t[ST_TIPR + SV_DES + k] = des[k]; t[ST_TIPR + SV_VWARM + k] = 0; }
    t[ST_TIPR + SV_HOLD] = 0.0;
    for (int k = 0; k < 7; k++) t[SE_ACT + k] = as.target[k];
    fl = as.arm.flags | ((as.arm.flags & PF_WARM_VALID) ? SKF_WARM_VALID : 0u);
    step = as.arm.step;
  } else if (live) {
    StackState ss;
    stack_load(state, flags, steps, stride, e, ss);
    double act[SK_ACT];
#pragma unroll
    for (int k = 0; k < SK_ACT; k++) act[k] = actions[(size_t)e * SK_ACT + k];
    bad = action_is_bad(actions + (size_t)e * SK_ACT, SK_ACT);      // integer test on the words in memory (see panda_step.h)
    if (bad) {    // NaN / Inf action: hold the current joints with an open gripper; the lane is flagged and terminated
#pragma unroll
      for (int k = 0; k < NARM; k++) act[k] = ss.arm.q[k];
      act[7] = 1.0;
    }
    for (int i = 0; i < SK_NV; i++) t[ST_X + i] = ss.warm[i];     // the warm start lives in the t area between the sub-steps
    float o[SK_OBS]; unsigned char dn = 0;
    open = stack_env_begin(g_stack_consts, ss, act, o, &dn, max_steps);
#pragma unroll
    for (int k = 0; k < SK_OBS; k++) obs[(size_t)SK_OBS * e + k] = o[k];      // observation and done flag are taken BEFORE the physics
    done[e] = dn;
    sk_state_to_lds(t, ss);
    for (int k = 0; k < NARM; k++) t[SE_ACT + k] = act[k];
    fl = ss.arm.flags; step = ss.arm.step;
  }
  const unsigned live_mask = (unsigned)(__ballot(live) & ((1ull << SK_LANES) - 1ull));
  sk_lds_double* const trig_lds = sm + 2 * ST_HEAD + SKW_TRIG + 2 * NARM * (live ? lane : 0);
  if (live) {      // sin / cos of the arm joints at the start of the step (then advanced with the joints)
    for (int k = 0; k < NARM; k++) { double sn, cs; sincos((double)t[SE_Q + k], &sn, &cs); trig_lds[k] = sn; trig_lds[NARM + k] = cs; }
  }
  __syncthreads();
#pragma clang loop unroll(disable)
  for (int s = 0; s < n_substeps; s++) {
#if defined(D3IL_SK_POISON)
    if (V == SKV_STACKING) {
      for (int q = lane; q < 2 * ST_HEAD + SKW_TRIG; q += WAVE) poison(q);
      if (live) {
        const int base = SKC_SHARED + lane * SE_SIZE - ST_HEAD;
        for (int q = ST_A0; q < ST_A0 + SK_NV; q++) poison(base + q);
        for (int q = ST_M; q < ST_BP; q++) poison(base + q);          // mass matrix, box rotations
        for (int q = ST_Z; q < ST_AUX + 2; q++) poison(base + q);     // joint axes / origins, finger axes, tip / hull poses, limit rows, aux, row size
        for (int q = SE_NCON; q < SE_END; q++) poison(base + q);      // contact count, need, records
      }
      __syncthreads();
    }
#endif
    if (live) sk_phase_pre<V>(t, g, fl, open, trig_lds);
    __syncthreads();
#if defined(D3IL_DEVICE_STATS)
    sk_collide_coop(g_stack_consts, sm, lane, live_mask, scratch + (size_t)blockIdx.x * SK_LANES * SG_SIZE);
#else
    sk_collide_coop(g_stack_consts, sm, lane, live_mask);
#endif
    if (live) sk_phase_mid<V>(t, g, fl);
    __syncthreads();
    // the solver takes two environments at a time, one per half wave (contacts in wrench form: no row area whose size could force a pair apart).
    // ONE call site: the solver is inlined once
#pragma clang loop unroll(disable)
    for (int e0 = 0; e0 < SK_LANES; e0 += 2) {
      const bool a0 = ((live_mask >> e0) & 1u) && sk_env_view(sm, e0)[SE_NEED] != 0.0;
      const bool a1 = ((live_mask >> (e0 + 1)) & 1u) && sk_env_view(sm, e0 + 1)[SE_NEED] != 0.0;
      if (!a0 && !a1) continue;
      const unsigned failed = sk_solve_dual(g_stack_consts, sm, e0, lane, a0, a1);
      if (((failed & 1u) && lane == e0) || ((failed & 2u) && lane == e0 + 1)) fl |= F_SOLVER_FAIL;
    }
    __syncthreads();
    if (live) sk_phase_post<V>(t, g, fl, trig_lds);
    __syncthreads();
  }
  if (!live) return;
#if defined(D3IL_DEVICE_STATS)
  {   // diagnostics build: the contact records of the last sub-step (count, then 8 doubles per contact) into the environment's scratch column
    const int nc = (int)t[SE_NCON];
    g[0] = (double)nc; g[1] = t[SE_NEED]; g[2] = 0;
    for (int i = 0; i < nc * SREC2; i++) g[8 + i] = t[SE_REC + i];
    for (int i = 0; i < 48; i++) g[500 + i] = t[ST_TIPR + i];
    for (int i = 0; i < 42; i++) g[560 + i] = t[ST_Z + i];
    for (int i = 0; i < 3 * SK_NV; i++) g[610 + i] = t[ST_X + i];      // x, a0, vel
    for (int i = 0; i < 45; i++) g[700 + i] = t[ST_M + i];
    for (int i = 0; i < 27; i++) g[750 + i] = t[ST_LIM + i];
    for (int i = 0; i < SE_REC - SE_Q; i++) g[780 + i] = t[SE_Q + i];
  }
#endif
  StackState ss;
  sk_state_from_lds(t, ss);
  ss.arm.flags = fl; ss.arm.step = step;
  if constexpr (V == SKV_ALIGNING) {
    AlignState as;
    as.arm = ss.arm; as.box = ss.box[0];
    for (int k = 0; k < 7; k++) as.target[k] = t[SE_ACT + k];
    double* sw = state + e + (size_t)AL_STATE_WARM * stride;
    for (int i = 0; i < 6; i++) sw[(size_t)i * stride] = t[ST_X + i];
    for (int i = 0; i < NDOF; i++) sw[(size_t)(6 + i) * stride] = t[ST_X + SK_ARM0 + i];
    if (reset) {
      for (int k = 0; k < NARM; k++) { as.arm.ikq[k] = 0; as.arm.ikqd[k] = 0; }
      as.arm.flags = (fl & ~(SKF_WARM_VALID | F_IK_VALID)) | PF_WARM_VALID;
      float o[AL_OBS];
      align_obs(as, o);
      align_store(state, flags, steps, stride, e, as);
#pragma unroll
      for (int k = 0; k < AL_OBS; k++) obs[(size_t)AL_OBS * e + k] = o[k];
      done[e] = 0; success[e] = 0; mode[e] = (unsigned short)(short)-1; info[e] = 0; info[(size_t)stride + e] = 0;
      return;
    }
    for (int k = 0; k < NARM; k++) { as.arm.ikq[k] = t[ST_TIPR + SV_IKQ + k]; as.arm.ikqd[k] = t[ST_TIPR + SV_IKQD + k]; }
    as.arm.flags = (fl & ~SKF_WARM_VALID) | F_IK_VALID | PF_WARM_VALID;
    if (bad) as.arm.flags |= F_SOLVER_FAIL | F_TERMINATED;
    double md = 0;
    align_step_end(g_align_task, as, &md);
    // pairs this engine does not evaluate: the box against the finger tips / hand (the rod is 30 cm long; only a policy that lowers the hand onto the box gets there)
    // (finger-tip boxes reach 15 mm below the TCP, the walls 93.5 mm above the body origin: contact from a height difference of 0.1085 m down; 6.5 mm guard)
    if (as.arm.tcp[2] - as.box.pos[2] < 0.115 && fabs(as.arm.tcp[0] - as.box.pos[0]) < 0.085 && fabs(as.arm.tcp[1] - as.box.pos[1]) < 0.085) as.arm.flags |= SKF_HAND_NEAR;      // walls reach 5.5 cm, tips 1 cm from the TCP axis; 2 cm guard
    align_store(state, flags, steps, stride, e, as);
    success[e] = (as.arm.flags & F_SUCCESS) ? 1 : 0;
    mode[e] = (unsigned short)(short)((int)((as.arm.flags & PF_MODE_MASK) >> PF_MODE_SHIFT) - 1);
    info[e] = md;
    return;
  }
  for (int i = 0; i < SK_NV; i++) ss.warm[i] = t[ST_X + i];
  if (reset) {
    float o[SK_OBS];
    stack_obs(ss, o);
    stack_store(state, flags, steps, stride, e, ss);
#pragma unroll
    for (int k = 0; k < SK_OBS; k++) obs[(size_t)SK_OBS * e + k] = o[k];
    done[e] = 0; success[e] = 0; mode[e] = 0; info[e] = 0;
    return;
  }
  double md = 0;
  stack_env_end(g_stack_consts, ss, &md);
  if (bad) ss.arm.flags |= F_SOLVER_FAIL | F_TERMINATED;
  stack_store(state, flags, steps, stride, e, ss);
  success[e] = (ss.arm.flags & F_SUCCESS) ? 1 : 0; mode[e] = (unsigned short)stack_mode_code(ss.arm.flags);
  info[e] = md;
}

__global__ __launch_bounds__(WAVE) void k_stacking_step(double* __restrict__ state, unsigned* __restrict__ flags, int* __restrict__ steps,
                                                        const double* __restrict__ actions, float* __restrict__ obs, unsigned char* __restrict__ done,
                                                        unsigned char* __restrict__ success, unsigned short* __restrict__ mode, double* __restrict__ info,
                                                        double* __restrict__ scratch, int n, int stride, int n_substeps, int max_steps,
                                                        const int reset, const unsigned char* __restrict__ reset_mask, const double* __restrict__ init_qpos,
                                                        const double* __restrict__ contexts) {
  coop_step_body<SKV_STACKING>(state, flags, steps, actions, obs, done, success, mode, info, scratch, n, stride, n_substeps, max_steps, reset, reset_mask, init_qpos, contexts);
}
// env.step() / env.reset() for the Aligning task on the wave-cooperative engine, variant 2: the rod robot and one free compound
// body of five box geoms; actions f64 [n][7] (the harness commands x, y and z), state layout D3IL_ALIGN_STATE_*, contexts f64 [n][14].
__global__ __launch_bounds__(WAVE) void k_aligning_step(double* __restrict__ state, unsigned* __restrict__ flags, int* __restrict__ steps,
                                                        const double* __restrict__ actions, float* __restrict__ obs, unsigned char* __restrict__ done,
                                                        unsigned char* __restrict__ success, unsigned short* __restrict__ mode, double* __restrict__ info,
                                                        double* __restrict__ scratch, int n, int stride, int n_substeps, int max_steps,
                                                        const int reset, const unsigned char* __restrict__ reset_mask, const double* __restrict__ init_qpos,
                                                        const double* __restrict__ contexts) {
  coop_step_body<SKV_ALIGNING>(state, flags, steps, actions, obs, done, success, mode, info, scratch, n, stride, n_substeps, max_steps, reset, reset_mask, init_qpos, contexts);
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
