// gen_kernels.h - HIP kernels of the Sorting task on the generic engine (included by rollout.hip).
//
// A workgroup owns GEN_LANES = 16 environments and runs three waves: the controller wave (open-loop IK chain, one lane per
// environment) and GEN_NSUB = 2 physics waves of eight environments each, in which every environment has a group of four lane
// PAIRS - pair l of the group owns cube l, pair 0 also the arm (gen_step.h).  The two lanes of a pair (sub-lanes, neighbours in
// the wave: lane = 2 (8 l + column) + sub) share the cube's contacts in the solver's contact loops (every other record each, sums
// exchanged by a DPP swap of neighbouring lanes) and hold identical copies of everything else; sub-lane 0 does the collision
// phases and the integration.  The physics waves keep each environment's vectors, matrices and the dense
// Newton Hessian in LDS (1073 doubles per environment, 134 KiB per workgroup), the contact records in the HBM scratch area
// and the cubes in the state buffer itself.  4096 environments are 256 workgroups: one per CU, three of its four SIMDs busy.
#pragma once
#include "gen_step.h"

namespace d3il {

constexpr int GEN_LDS_H = GL_SIZE * GEN_LANES * 8;   // 1073 x 16 doubles = 134 KiB
constexpr int GEN_LDS_X = 2 * 2 * NARM * GEN_LANES * 8;
constexpr int GEN_LDS_STEP = GEN_LDS_H + GEN_LDS_X;

__device__ __forceinline__ void gen_load_arm(const double* __restrict__ state, const unsigned* __restrict__ flags, const int* __restrict__ steps, int stride, int e,
                                             EnvState& st, bool with_ik) {
  const double* s = state + e;
  for (int i = 0; i < NDOF; i++) st.q[i] = s[(D3IL_STATE_QPOS + i) * (size_t)stride];
  for (int i = 0; i < NDOF; i++) st.v[i] = s[(D3IL_STATE_QVEL + i) * (size_t)stride];
  for (int i = 0; i < NARM; i++) st.bias[i] = s[(D3IL_STATE_BIAS + i) * (size_t)stride];
  for (int i = 0; i < 3; i++) st.tcp[i] = s[(D3IL_STATE_TCP + i) * (size_t)stride];
  if (with_ik) {
    for (int i = 0; i < NARM; i++) st.ikq[i] = s[(D3IL_STATE_IK_Q + i) * (size_t)stride];
    for (int i = 0; i < NARM; i++) st.ikqd[i] = s[(D3IL_STATE_IK_QD + i) * (size_t)stride];
  }
  st.flags = flags[e]; st.step = steps[e];
}
__device__ __forceinline__ void gen_store_arm(double* __restrict__ state, unsigned* __restrict__ flags, int* __restrict__ steps, int stride, int e, const EnvState& st,
                                              bool with_ik) {
  double* s = state + e;
  for (int i = 0; i < NDOF; i++) s[(D3IL_STATE_QPOS + i) * (size_t)stride] = st.q[i];
  for (int i = 0; i < NDOF; i++) s[(D3IL_STATE_QVEL + i) * (size_t)stride] = st.v[i];
  for (int i = 0; i < NARM; i++) s[(D3IL_STATE_BIAS + i) * (size_t)stride] = st.bias[i];
  for (int i = 0; i < 3; i++) s[(D3IL_STATE_TCP + i) * (size_t)stride] = st.tcp[i];
  if (with_ik) {
    for (int i = 0; i < NARM; i++) s[(D3IL_STATE_IK_Q + i) * (size_t)stride] = st.ikq[i];
    for (int i = 0; i < NARM; i++) s[(D3IL_STATE_IK_QD + i) * (size_t)stride] = st.ikqd[i];
  }
  flags[e] = st.flags; steps[e] = st.step;
}

#if defined(D3IL_DEVICE_STATS) && defined(D3IL_STATS_PER_WAVE)
#define GEN_BARRIER_TIMED(w) do { const unsigned long long tb_ = wall_clock64(); __syncthreads(); \
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 1024) atomicAdd(&d3il::g_dev_wave[blockIdx.x * 4 + 3][w], wall_clock64() - tb_); } while (0)
#else
#define GEN_BARRIER_TIMED(w) __syncthreads()
#endif

// The arm's state lives in the t area (LDS) between the phases: the phases that need it (1: dynamics, 3b: q, 5: integration) read it from there and write back
// what they change.  Held in registers across the sub-step loop (56 + registers of the arm lanes' wave) it was stored to and reloaded from scratch around every
// out-of-line phase - several hundred scratch instructions per wave and sub-step (round 6).
__device__ __forceinline__ void gen_park_arm(const PushScratch sc, const EnvState& st) {
#pragma unroll
  for (int i = 0; i < NDOF; i++) { GLS(GL_ARMST + i) = st.q[i]; GLS(GL_ARMST + NDOF + i) = st.v[i]; }
#pragma unroll
  for (int i = 0; i < NARM; i++) GLS(GL_ARMST + 2 * NDOF + i) = st.bias[i];
#pragma unroll
  for (int i = 0; i < 3; i++) GLS(GL_ARMST + 2 * NDOF + NARM + i) = st.tcp[i];
  GLS(GL_ARMST + 28) = (double)st.flags; GLS(GL_ARMST + 29) = (double)st.step;
}
__device__ __forceinline__ void gen_unpark_arm(const PushScratch sc, EnvState& st) {
#pragma unroll
  for (int i = 0; i < NDOF; i++) { st.q[i] = GLS(GL_ARMST + i); st.v[i] = GLS(GL_ARMST + NDOF + i); }
#pragma unroll
  for (int i = 0; i < NARM; i++) st.bias[i] = GLS(GL_ARMST + 2 * NDOF + i);
#pragma unroll
  for (int i = 0; i < 3; i++) st.tcp[i] = GLS(GL_ARMST + 2 * NDOF + NARM + i);
  st.flags = (unsigned)GLS(GL_ARMST + 28); st.step = (int)GLS(GL_ARMST + 29);
}

// The physics role of the step kernel (one wave; GEN_NSUB of them per workgroup).  Built into the kernel by default.  -DD3IL_GT_INLINE makes it a function of
// its own - called ONCE per env step, with the tree solver built INTO it: the solver's callee-saved register block (84 KB of scratch stores per wave and call,
// profiles/r05/README.md) is then saved once per step instead of once per sub-step, and the role gets a register allocation of its own.
#if defined(D3IL_GT_INLINE)
#define D3IL_GEN_ROLE_ATTR __device__ __attribute__((noinline))
#else
#define D3IL_GEN_ROLE_ATTR __device__ __forceinline__
#endif
template <bool FAST, bool RS>
D3IL_GEN_ROLE_ATTR void gen_physics_role(double* __restrict__ state, unsigned* __restrict__ flags, int* __restrict__ steps, const double* __restrict__ actions,
                                         float* __restrict__ obs, unsigned char* __restrict__ done, unsigned char* __restrict__ success, unsigned short* __restrict__ mode,
                                         double* __restrict__ scratch, int n, int stride, int n_substeps, int max_steps, double* tbl,
                                         double (*xch)[2 * NARM][GEN_LANES], int lane, int role) {
  const PandaConsts& c = kAvoidingConsts;
  const GenConsts& gc = g_gen_consts;
  constexpr int CPW = GEN_LANES / GEN_NSUB;                      // environment columns per physics wave
  const int sub = lane & (GEN_NSUB - 1), pr = lane / GEN_NSUB;   // sub-lane of the pair, pair index in the wave
  const int col = (role - 1) * CPW + (pr & (CPW - 1)), l = pr / CPW;   // pair l of environment column col
  const int e = blockIdx.x * GEN_LANES + col;
  const bool slive = e < n && l < gc.nb;                          // the lane takes part in the solver
  const bool plive = slive && sub == 0;                           // ... and in the per-cube phases
  const bool arm_lane = plive && l == 0;
  const size_t ei = e < n ? e : 0;
  PushScratch sc{(push_lds_double*)(tbl + col * GL_SIZE), (push_glb_double*)(scratch + (size_t)blockIdx.x * GG_BLOCK * GEN_LANES + 2 * col), GEN_LANES, (push_glb_double*)(state + (size_t)42 * stride + ei), stride};
  unsigned lfl = 0;
  bool warm_valid = false;
  const double grav[3] = {c.gravity[0], c.gravity[1], c.gravity[2]};
  if (slive) warm_valid = (flags[e] & PF_WARM_VALID) != 0;
  if (arm_lane) {      // observation and `done` BEFORE the physics (gym_env_wrapper.py:88-93): written out at once, the arm's state goes to the t area
    EnvState st;
    float o[GEN_SORT_OBS]; unsigned char dn = 0;
    gen_load_arm(state, flags, steps, stride, e, st, false);
    sort_step_begin(gc, st, sc, o, &dn, max_steps);
    const int od = 2 + 3 * gc.nb;
    for (int k = 0; k < od; k++) obs[(size_t)od * e + k] = o[k];
    done[e] = dn;
    gen_park_arm(sc, st);
  }
#pragma clang loop unroll(disable)
  for (int s = 0; s < n_substeps; s++) {
    GEN_BARRIER_TIMED(role);
    PUSH_TIC;
    if (arm_lane) {
      const int b = s & 1;
      double qd[NARM], qdd[NARM], tau[NARM], ff[NFING];
#pragma unroll
      for (int k = 0; k < NARM; k++) { qd[k] = xch[b][k][col]; qdd[k] = xch[b][NARM + k][col]; }
      EnvState st;
      gen_unpark_arm(sc, st);
      push_control(c, st, qd, qdd, 0.04, false, tau, ff);
      gen_phase1(c, gc, st, sc, tau, ff);
      gen_park_arm(sc, st);      // (phase 1 refreshes bias, tcp and may raise a flag)
    }
    PUSH_TOC(0);
    int cnt = 0;
    if (slive) cnt = gen_phase2(gc, sc, l, grav, lfl, GEN_NSUB > 1 ? sub : -1);      // both sub-lanes: identical work, each stores the record fields of its parity (gen_put)
    gen_sync();
    PUSH_TOC(1);
    if (slive) gen_phase3(gc, sc, l, cnt, c.rod_r, c.rod_h, lfl, GEN_NSUB > 1 ? sub : -1);
    if (RS && plive) gen_phase3r(gc, sc, l, gc.nb, c.rod_r, c.rod_h, lfl);
    gen_sync();
    if (arm_lane) { EnvState st; gen_unpark_arm(sc, st); gen_phase3b<RS>(c, gc, st, sc, gc.nb, lfl); gen_arm_reduce<RS>(gc, sc, warm_valid); }
    gen_sync();
    PUSH_TOC(2);
    // environments without cube <-> cube and rod contacts (every cube on static boxes only) solve their cubes one by one in the kernel's own registers; a wave
    // none of whose environments has a coupled island never calls the tree solver (and does not pay for its register save block).
    // The choice is per WAVE: a wave with one coupled environment sends all of its environments through the tree solver.  An environment's result must
    // not depend on who shares its wave (tests/test_gpu_permutation.py), so the two solvers have to agree BIT FOR BIT on a one-node island: both are written
    // over the same helpers in the same order and gen_tree.h is compiled with expression-level contraction (`fp contract(on)`), which leaves the compiler no
    // freedom to fuse across statements differently in the two bodies (with hipcc's default, contract(fast), 3707 of 4096 environments depended on their
    // position; with it, none - tools/gpu_perm_push_sort.py, profiles/r05/lone_solver/).  -DD3IL_LONE_PER_ENV: the choice per environment (a wave with
    // both kinds runs both solvers one after the other: -7 % in the all-contact regime of Sorting, -11 % of Pushing).
#if defined(D3IL_LONE_PER_ENV)
    const bool lone_env = gen_uncoupled(gc, sc);
#else
    const bool lone_env = !wave_any(slive && !gen_uncoupled(gc, sc));
#endif
    if (slive && lone_env) lfl |= gen_lone_solve<GEN_NSUB>(gc, sc, l, warm_valid, sub);
    if (slive && !lone_env) lfl |= gen_tree_solve<1, GEN_NSUB>(gc, sc, l, warm_valid, sub);
    gen_sync();
    PUSH_TOC(8);
    // The joint solver (islands that are not trees: the rod on two cubes, three cubes touching each other, an arm joint at its limit) by ALL 64 lanes of the wave,
    // one environment after the other: such an island keeps its environment in this path for tens of sub-steps, and with the eight lanes of its own group it was
    // what the slowest wave of a launch spent half its time in (round 6, profiles/r06/sort_phases_per_wave.log).  Contacts, Hessian block rows and vector entries are
    // dealt out over 64 lanes instead of eight; the other environments of the wave have nothing to do meanwhile anyway.  Every environment goes through the same
    // 64-lane code whoever shares its wave: results do not depend on the batch position (tests/test_gpu_permutation.py).
    {
      const bool need = slive && gen_joint_work<RS>(gc, sc);
      unsigned long long todo = __ballot(need && l == 0 && sub == 0);
      while (todo) {
        const int src = __builtin_ctzll(todo);
        todo &= todo - 1;
        const int col0 = (role - 1) * CPW + ((src / GEN_NSUB) & (CPW - 1));
        const size_t e0 = (size_t)blockIdx.x * GEN_LANES + col0;
        const PushScratch sc0{(push_lds_double*)(tbl + col0 * GL_SIZE), (push_glb_double*)(scratch + (size_t)blockIdx.x * GG_BLOCK * GEN_LANES + 2 * col0), GEN_LANES,
                              (push_glb_double*)(state + (size_t)42 * stride + e0), stride};
        const bool wv0 = __shfl((int)warm_valid, src) != 0;
        unsigned f0 = 0;
        gen_phase4_multi<RS>(gc, sc0, lane, WAVE, wv0, f0);
        if (col == col0) lfl |= f0;
        gen_sync();
      }
    }
    gen_sync();
    PUSH_TOC(9);
    if (arm_lane) { EnvState st; gen_unpark_arm(sc, st); gen_phase5_arm(c, gc, st, sc); gen_park_arm(sc, st); }
    if (plive) gen_phase5_cube(gc, sc, l, c.timestep);
    gen_sync();
    warm_valid = true;
  }
  if (plive && l > 0) GLS(GL_INFO + 4 + l) = (double)lfl;
  gen_sync();
  if (arm_lane) {
    int code = 0;
    EnvState st;
    gen_unpark_arm(sc, st);
    for (int k = 1; k < gc.nb; k++) lfl |= (unsigned)GLS(GL_INFO + 4 + k);
    st.flags |= F_IK_VALID | PF_WARM_VALID | lfl;
    if (action_is_bad(actions + (size_t)e * 7)) st.flags |= F_SOLVER_FAIL | F_TERMINATED;
    sort_step_end(gc, st, sc, &code);
    gen_store_arm(state, flags, steps, stride, e, st, false);
    success[e] = (st.flags & F_SUCCESS) ? 1 : 0; mode[e] = (unsigned short)code;
  }
}

// env.step() for the Sorting task (RS = false) and the Inserting task (RS = true: the engine with contacts of the arm block, gen_step.h)
template <bool FAST, bool RS>
__global__ __launch_bounds__((1 + GEN_NSUB) * WAVE) void k_sorting_step(double* __restrict__ state, unsigned* __restrict__ flags,
                                                           int* __restrict__ steps, const double* __restrict__ actions, float* __restrict__ obs,
                                                           unsigned char* __restrict__ done, unsigned char* __restrict__ success, unsigned short* __restrict__ mode,
                                                           double* __restrict__ scratch, int n, int stride, int n_substeps, int max_steps) {
  extern __shared__ double smem[];
  double* tbl = smem;                                    // [GEN_LANES][GL_SIZE]
  double (*xch)[2 * NARM][GEN_LANES] = (double (*)[2 * NARM][GEN_LANES])(smem + GL_SIZE * GEN_LANES);
  const int lane = threadIdx.x & (WAVE - 1);
  const int role = threadIdx.x / WAVE;
  const PandaConsts& c = kAvoidingConsts;                // the arm is the Avoiding arm (same robot XML / gin / URDF)
  const GenConsts& gc = g_gen_consts;
  if (role == 0) {
    const int e = blockIdx.x * GEN_LANES + lane;
    const bool live = lane < GEN_LANES && e < n;         // the other lanes only take part in the barriers
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
      GEN_BARRIER_TIMED(0);
    }
    if (live) {
      double* so = state + e;
#pragma unroll
      for (int i = 0; i < NARM; i++) { so[(D3IL_STATE_IK_Q + i) * (size_t)stride] = ikq[i]; so[(D3IL_STATE_IK_QD + i) * (size_t)stride] = ikqd[i]; }
    }
  } else {
    gen_physics_role<FAST, RS>(state, flags, steps, actions, obs, done, success, mode, scratch, n, stride, n_substeps, max_steps, tbl, xch, lane, role);
  }
}

// env.reset(random=False, context) for masked environments; contexts: f64 [n][7 nb] = nb x (pos3, quat4), red boxes first
__global__ __launch_bounds__(WAVE) void k_sorting_reset(const double* __restrict__ init_qpos, const unsigned char* __restrict__ mask,
                                                        const double* __restrict__ contexts, double* __restrict__ state, unsigned* __restrict__ flags,
                                                        int* __restrict__ steps, float* __restrict__ obs, unsigned char* __restrict__ done,
                                                        unsigned char* __restrict__ success, unsigned short* __restrict__ mode, double* __restrict__ scratch, int n, int stride) {
  extern __shared__ double smem[];
  const int lane = threadIdx.x;
  const int e = blockIdx.x * GEN_LANES + lane;
  if (lane >= GEN_LANES || e >= n) return;
  if (mask && !mask[e]) return;
  const GenConsts& gc = g_gen_consts;
  EnvState st;
  double iq[NARM];
#pragma unroll
  for (int k = 0; k < NARM; k++) iq[k] = init_qpos[k];
  PushScratch sc{(push_lds_double*)(smem + lane * GL_SIZE), (push_glb_double*)(scratch + (size_t)blockIdx.x * GG_BLOCK * GEN_LANES + 2 * lane), GEN_LANES, (push_glb_double*)(state + (size_t)42 * stride + e), stride};
  float o[GEN_SORT_OBS];
  st.flags = 0; st.step = 0;
  gen_env_reset(kAvoidingConsts, gc, st, sc, iq, contexts + (size_t)e * 7 * gc.nb, o);
  gen_store_arm(state, flags, steps, stride, e, st, true);
  const int od = 2 + 3 * gc.nb;
  for (int k = 0; k < od; k++) obs[(size_t)od * e + k] = o[k];
  int code = 0;
  if (gc.task == GEN_TASK_SORTING) for (int i = 0; i < gc.nb; i++) code |= 1 << (7 - i);        // np.packbits of the all -1 mode vector (Inserting: no letters yet)
  if (gc.task == GEN_TASK_PUSHING) code = -1;                                                    // no mode yet (int16 -1, pushing.py:341)
  done[e] = 0; success[e] = 0; mode[e] = (unsigned short)code;
}

}  // namespace d3il
