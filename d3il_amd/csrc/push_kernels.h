// push_kernels.h - HIP kernels of the Pushing task (included by rollout.hip).
//
// Same execution shape as the Avoiding step: one environment per lane, two cooperating waves per workgroup (the
// controller wave runs the open-loop IK chain, the physics wave runs dynamics + collision + constraint solve +
// integration, one workgroup barrier per sub-step, set-points handed over through an LDS slot).  A workgroup owns
// PUSH_LANES = 24 environments: the coupled constraint solver keeps its per-contact table, the arm mass matrix, the cube
// Hessian and the elimination matrices of each environment in LDS (710 doubles per environment, lane-strided =>
// conflict-free ds_read/write_b64), 133 KiB per workgroup, i.e. one workgroup per CU; 4096 environments are 171
// workgroups on 171 of the 256 CUs.  The rarely used memory-resident solver (arm joint at a limit, rod on both cubes) works in an HBM scratch area.
#pragma once
#include "push_step.h"

namespace d3il {

constexpr int PUSH_LDS_H = PT_SIZE * PUSH_LANES * 8;             // coupled-solver table
constexpr int PUSH_LDS_X = 2 * 2 * NARM * PUSH_LANES * 8;        // set-point exchange (double buffered)
constexpr int PUSH_LDS_STEP = PUSH_LDS_H + PUSH_LDS_X;
constexpr int PUSH_OBS = 8;

__device__ __forceinline__ void push_load(const double* __restrict__ state, const unsigned* __restrict__ flags, const int* __restrict__ steps,
                                          int stride, int e, PushState& ps, bool with_ik) {
  const double* s = state + e;
  EnvState& st = ps.arm;
  for (int i = 0; i < NDOF; i++) st.q[i] = s[(D3IL_STATE_QPOS + i) * (size_t)stride];
  for (int i = 0; i < NDOF; i++) st.v[i] = s[(D3IL_STATE_QVEL + i) * (size_t)stride];
  for (int i = 0; i < NARM; i++) st.bias[i] = s[(D3IL_STATE_BIAS + i) * (size_t)stride];
  for (int i = 0; i < 3; i++) st.tcp[i] = s[(D3IL_STATE_TCP + i) * (size_t)stride];
  if (with_ik) {
    for (int i = 0; i < NARM; i++) st.ikq[i] = s[(D3IL_STATE_IK_Q + i) * (size_t)stride];
    for (int i = 0; i < NARM; i++) st.ikqd[i] = s[(D3IL_STATE_IK_QD + i) * (size_t)stride];
  }
  int k = PUSH_STATE_BOX;
  for (int b = 0; b < PUSH_NB; b++) {
    for (int i = 0; i < 3; i++) ps.box[b].pos[i] = s[(size_t)(k++) * stride];
    for (int i = 0; i < 4; i++) ps.box[b].quat[i] = s[(size_t)(k++) * stride];
    for (int i = 0; i < 6; i++) ps.box[b].vel[i] = s[(size_t)(k++) * stride];
  }
  st.flags = flags[e]; st.step = steps[e];
}
__device__ __forceinline__ void push_store(double* __restrict__ state, unsigned* __restrict__ flags, int* __restrict__ steps, int stride, int e,
                                           const PushState& ps, bool with_ik) {
  double* s = state + e;
  const EnvState& st = ps.arm;
  for (int i = 0; i < NDOF; i++) s[(D3IL_STATE_QPOS + i) * (size_t)stride] = st.q[i];
  for (int i = 0; i < NDOF; i++) s[(D3IL_STATE_QVEL + i) * (size_t)stride] = st.v[i];
  for (int i = 0; i < NARM; i++) s[(D3IL_STATE_BIAS + i) * (size_t)stride] = st.bias[i];
  for (int i = 0; i < 3; i++) s[(D3IL_STATE_TCP + i) * (size_t)stride] = st.tcp[i];
  if (with_ik) {
    for (int i = 0; i < NARM; i++) s[(D3IL_STATE_IK_Q + i) * (size_t)stride] = st.ikq[i];
    for (int i = 0; i < NARM; i++) s[(D3IL_STATE_IK_QD + i) * (size_t)stride] = st.ikqd[i];
  }
  int k = PUSH_STATE_BOX;
  for (int b = 0; b < PUSH_NB; b++) {
    for (int i = 0; i < 3; i++) s[(size_t)(k++) * stride] = ps.box[b].pos[i];
    for (int i = 0; i < 4; i++) s[(size_t)(k++) * stride] = ps.box[b].quat[i];
    for (int i = 0; i < 6; i++) s[(size_t)(k++) * stride] = ps.box[b].vel[i];
  }
  flags[e] = st.flags; steps[e] = st.step;
}
__device__ __forceinline__ void push_store_outputs(const PushState& ps, int e, int stride, const float* o, unsigned char dn, double reward, double mean_distance,
                                                   float* __restrict__ obs, unsigned char* __restrict__ done, unsigned char* __restrict__ success,
                                                   unsigned short* __restrict__ mode, double* __restrict__ info) {
  for (int k = 0; k < PUSH_OBS; k++) obs[(size_t)PUSH_OBS * e + k] = o[k];
  done[e] = dn; success[e] = (ps.arm.flags & F_SUCCESS) ? 1 : 0;
  mode[e] = (unsigned short)(short)((int)((ps.arm.flags & PF_MODE_MASK) >> PF_MODE_SHIFT) - 1);   // int16: -1 .. 3
  info[e] = mean_distance; info[(size_t)stride + e] = reward;
}

// env.step() for the Pushing task
template <bool FAST>
__global__ __launch_bounds__(2 * WAVE) void k_pushing_step_split(double* __restrict__ state, unsigned* __restrict__ flags, int* __restrict__ steps,
                                                                 const double* __restrict__ actions, float* __restrict__ obs, unsigned char* __restrict__ done,
                                                                 unsigned char* __restrict__ success, unsigned short* __restrict__ mode, double* __restrict__ info,
                                                                 double* __restrict__ scratch, int n, int stride, int n_substeps, int max_steps) {
  extern __shared__ double smem[];
  double* tbl = smem;                                    // [PT_SIZE][PUSH_LANES]
  double (*xch)[2 * NARM][PUSH_LANES] = (double (*)[2 * NARM][PUSH_LANES])(smem + PT_SIZE * PUSH_LANES);
  const int lane = threadIdx.x & (WAVE - 1);
  const int role = threadIdx.x / WAVE;
  const int e = blockIdx.x * PUSH_LANES + lane;
  const bool live = lane < PUSH_LANES && e < n;          // the other lanes only take part in the barriers
  const PandaConsts& c = kAvoidingConsts;                // the arm is the Avoiding arm (same robot XML / gin / URDF)
  const PushConsts& pc = g_push_consts;
  if (role == 0) {
    double ikq[NARM], ikqd[NARM], q0[NARM], des[7];
    unsigned fl = 0;
    double vwarm[7];
    vwarm[6] = 0.0;
    if (live) {
      const double* sp = state + e;
      double act[7];
#pragma unroll
      for (int i = 0; i < NARM; i++) {
        ikq[i] = sp[(D3IL_STATE_IK_Q + i) * (size_t)stride]; ikqd[i] = sp[(D3IL_STATE_IK_QD + i) * (size_t)stride];
        q0[i] = sp[(D3IL_STATE_QPOS + i) * (size_t)stride];
      }
#pragma unroll
      for (int k = 0; k < 7; k++) act[k] = actions[(size_t)e * 7 + k];
      fl = flags[e];
      sanitize_action(act, actions + (size_t)e * 7);
      make_setpoint(act, des);
    }
#pragma clang loop unroll(disable)
    for (int s = 0; s < n_substeps; s++) {
      if (live) {
        ik_update<FAST>(c, des, des + 3, q0, fl, ikq, ikqd, vwarm);
        const int b = s & 1;
#pragma unroll
        for (int k = 0; k < NARM; k++) { xch[b][k][lane] = ikq[k]; xch[b][NARM + k][lane] = ikqd[k]; }
      }
      __syncthreads();
    }
    if (live) {
      double* so = state + e;
#pragma unroll
      for (int i = 0; i < NARM; i++) { so[(D3IL_STATE_IK_Q + i) * (size_t)stride] = ikq[i]; so[(D3IL_STATE_IK_QD + i) * (size_t)stride] = ikqd[i]; }
    }
  } else {
    // physics wave.  Lanes [0, PUSH_LANES): arm + cube 0 of environment e; lanes [PUSH_LANES, 2 PUSH_LANES): cube 1 of the
    // same environments (same table column), so the two decoupled cube solves of a sub-step run side by side.  The pair
    // exchanges cube states, warm starts, the "jointly solved" decision and its result through the LDS table; both
    // lanes are in the same wave, so program order + a wavefront fence is all the synchronisation needed.
    const int col = lane < PUSH_LANES ? lane : lane - PUSH_LANES;
    const int cube_id = lane < PUSH_LANES ? 0 : 1;
    const int ee = blockIdx.x * PUSH_LANES + col;
    const bool plive = lane < 2 * PUSH_LANES && ee < n;
    const bool arm_lane = plive && cube_id == 0;
    const size_t ei = plive ? ee : 0;
    PushScratch sc{(push_lds_double*)(tbl + col), (push_glb_double*)(scratch + ei), stride, (push_glb_double*)(state + (size_t)PUSH_STATE_WARM * stride + ei), stride};
    EnvState st;
    BoxState own, other;
    double warm6[6], owarm[6];
    float o[PUSH_OBS]; unsigned char dn = 0; double reward = 0, mean_distance = 0;
    unsigned pflags = 0;
    bool warm_valid = false;
    const int crow = PUSH_STATE_BOX + 13 * cube_id, pub = PT_PAIR + 19 * cube_id, opub = PT_PAIR + 19;
    auto publish = [&]() {
      for (int k = 0; k < 3; k++) PTS(pub + k) = own.pos[k];
      for (int k = 0; k < 4; k++) PTS(pub + 3 + k) = own.quat[k];
      for (int k = 0; k < 6; k++) { PTS(pub + 7 + k) = own.vel[k]; PTS(pub + 13 + k) = warm6[k]; }
    };
    auto read_other = [&]() {
      for (int k = 0; k < 3; k++) other.pos[k] = PTS(opub + k);
      for (int k = 0; k < 4; k++) other.quat[k] = PTS(opub + 3 + k);
      for (int k = 0; k < 6; k++) { other.vel[k] = PTS(opub + 7 + k); owarm[k] = PTS(opub + 13 + k); }
    };
    auto pair_sync = []() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); };
    if (plive) {
      const double* s = state + ee;
      for (int i = 0; i < 3; i++) own.pos[i] = s[(size_t)(crow + i) * stride];
      for (int i = 0; i < 4; i++) own.quat[i] = s[(size_t)(crow + 3 + i) * stride];
      for (int i = 0; i < 6; i++) own.vel[i] = s[(size_t)(crow + 7 + i) * stride];
      for (int i = 0; i < 6; i++) warm6[i] = s[(size_t)(PUSH_STATE_WARM + 6 * cube_id + i) * stride];
      warm_valid = (flags[ee] & PF_WARM_VALID) != 0;
      publish();
    }
    pair_sync();
    if (arm_lane) {
      PushState ps;
      push_load(state, flags, steps, stride, ee, ps, false);
      read_other();
      ps.box[0] = own; ps.box[1] = other;
      push_step_begin(pc, ps, o, &reward, &dn, max_steps);
      st = ps.arm;
    }
#pragma clang loop unroll(disable)
    for (int s = 0; s < n_substeps; s++) {
      __syncthreads();
      if (arm_lane) {
        const int b = s & 1;
        double qd[NARM], qdd[NARM], tau[NARM], ff[NFING], cw[12];
        BoxState box[2];
#pragma unroll
        for (int k = 0; k < NARM; k++) { qd[k] = xch[b][k][col]; qdd[k] = xch[b][NARM + k][col]; }
        read_other();
        box[0] = own; box[1] = other;
#pragma unroll
        for (int k = 0; k < 6; k++) { cw[k] = warm6[k]; cw[6 + k] = owarm[k]; }
        push_control(c, st, qd, qdd, 0.04, false, tau, ff);
        const int solved_mask = push_substep_arm(c, pc, st, box, cw, sc, tau, ff);
        PTS(PT_SOLVED) = (double)solved_mask;
      }
      pair_sync();
      if (plive) {
        const bool solved = (((int)PTS(PT_SOLVED) >> cube_id) & 1) != 0;
        const double grav[3] = {c.gravity[0], c.gravity[1], c.gravity[2]};
        push_substep_cube(pc, grav, c.timestep, own, warm6, cube_id, solved, warm_valid, pflags, sc);
        warm_valid = true;
        publish();
      }
      pair_sync();
    }
    if (plive && cube_id == 1) PTS(PT_PFLAG) = (double)pflags;
    pair_sync();
    if (arm_lane) {
      read_other();
      PushState ps;
      ps.arm = st; ps.box[0] = own; ps.box[1] = other;
      ps.arm.flags |= F_IK_VALID | PF_WARM_VALID | pflags | (unsigned)PTS(PT_PFLAG);
      if (action_is_bad(actions + (size_t)ee * 7)) ps.arm.flags |= F_SOLVER_FAIL | F_TERMINATED;
      push_step_end(pc, ps, &mean_distance);
      push_store(state, flags, steps, stride, ee, ps, false);
      push_store_outputs(ps, ee, stride, o, dn, reward, mean_distance, obs, done, success, mode, info);
    }
    if (plive && cube_id == 1) {      // cube 1 rows and warm start are written by their owner (cube 0 went through push_store)
      double* so = state + ee;
      for (int i = 0; i < 3; i++) so[(size_t)(crow + i) * stride] = own.pos[i];
      for (int i = 0; i < 4; i++) so[(size_t)(crow + 3 + i) * stride] = own.quat[i];
      for (int i = 0; i < 6; i++) so[(size_t)(crow + 7 + i) * stride] = own.vel[i];
    }
    if (plive) {
      double* so = state + ee;
      for (int i = 0; i < 6; i++) so[(size_t)(PUSH_STATE_WARM + 6 * cube_id + i) * stride] = warm6[i];
    }
  }
}

// env.reset(random=False, context) for masked environments; contexts: f64 [n][14] = 2 x (pos3, quat4)
__global__ __launch_bounds__(WAVE) void k_pushing_reset(const double* __restrict__ init_qpos, const unsigned char* __restrict__ mask,
                                                        const double* __restrict__ contexts, double* __restrict__ state, unsigned* __restrict__ flags,
                                                        int* __restrict__ steps, float* __restrict__ obs, unsigned char* __restrict__ done,
                                                        unsigned char* __restrict__ success, unsigned short* __restrict__ mode, double* __restrict__ info,
                                                        double* __restrict__ scratch, int n, int stride) {
  extern __shared__ double smem[];
  const int lane = threadIdx.x;
  const int e = blockIdx.x * PUSH_LANES + lane;
  if (lane >= PUSH_LANES || e >= n) return;
  if (mask && !mask[e]) return;
  const PushConsts& pc = g_push_consts;
  PushState ps;
  double iq[NARM], ctx[14];
#pragma unroll
  for (int k = 0; k < NARM; k++) iq[k] = init_qpos[k];
  for (int k = 0; k < 14; k++) ctx[k] = contexts[(size_t)e * 14 + k];
  PushScratch sc{(push_lds_double*)(smem + lane), (push_glb_double*)(scratch + e), stride, (push_glb_double*)(state + (size_t)PUSH_STATE_WARM * stride + e), stride};
  float o[PUSH_OBS];
  ps.arm.flags = 0; ps.arm.step = 0;
  push_env_reset(kAvoidingConsts, pc, ps, sc, iq, ctx, o);
  push_store(state, flags, steps, stride, e, ps, true);
  push_store_outputs(ps, e, stride, o, 0, 0.0, 0.0, obs, done, success, mode, info);
}

}  // namespace d3il
