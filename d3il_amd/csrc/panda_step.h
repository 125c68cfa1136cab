// panda_step.h - per-environment math of the fused Avoiding step (one environment per lane).
//
// Everything here is __host__ __device__ and fully unrollable: in the HIP kernels each lane keeps
// its environment in registers, the constant block (PandaConsts) arrives through scalar loads.
// The same functions are compiled for the host ONLY to (a) finish the constant block at
// d3il_create (inverse weights at qpos0) and (b) by tests/hostcheck, which is not part of the
// product path.
//
// Formulation (deliberately different from the oracle's world-frame, MuJoCo-style restatement):
// link-frame recursive Newton-Euler for the bias forces, link-frame composite-rigid-body for the
// mass matrix, packed LDL^T, primal Newton on the soft-constraint problem with an exact 1-D line
// search.  Reference path being replaced, per physics sub-step:
//   MjRobot.prepare_step  (sims/mj_beta/MjRobot.py:125-131)
//     CartPosQuatImpedenceController.getControl (controllers/IKControllers.py:163-323)
//     JointPDController.getControl              (controllers/Controller.py:164-185)
//     RobotBase.fing_ctrl_step/preprocessCommand (core/Robots.py:441-476,530-572)
//   mujoco.mj_step        (sims/mj_beta/MjScene.py:110-111)  [ext: MuJoCo 2.3.2]
//   MjRobot.receiveState  (sims/mj_beta/MjRobot.py:133-184)
#pragma once
#include "panda_consts.h"

#include <type_traits>

// Rare, register-hungry paths (Jacobi eigen-solver, general constraint Newton) are kept out of line so that the hot
// path of the fused kernel stays in registers; they take their operands through private memory.
// not_tail_called (round 6): the AMDGPU calling convention makes v40.., a32.. callee-saved; a device function that needs the whole register file (the
// tree solver: 256 + 246) stored and reloaded 328 of them around EVERY call - 84 KB of scratch per wave and sub-step, the source of the generic engine's
// HBM traffic (x 270 - 900 the algorithmic bytes in round 5).  LLVM drops the callee-saved convention for an internal, non-recursive function under
// inter-procedural register allocation (TargetFrameLowering::isSafeForNoCSROpt; on by default for this target) unless one of its call sites carries the `tail`
// marker - which the optimiser puts on every call that cannot see the caller's stack.  `not_tail_called` keeps the marker off: no save block, the caller
// keeps what IT has live across the call (little, by construction of the phases).
#if defined(__HIPCC__)
#define D3IL_NOINLINE __host__ __device__ __attribute__((noinline, not_tail_called))
#else
#define D3IL_NOINLINE __attribute__((noinline))
#endif

// D3IL_RARE: the contact path (make_rod_contact, solve_contact5).  Same-box A/B on MI355X (bench.py, 4096 envs, full
// episodes, device flags of d3il_amd/build.py): inlined 2.33 M env-steps/s, out of line 2.13 M.  NOTE: an out-of-line
// build WITHOUT the finite-math device flags was observed to be miscompiled by hipcc 7.2 (one state register, qpos[1],
// corrupted at the end of the step; every other value exact) - the GPU suite's golden-rollout tests catch this class
// of failure; keep them green for any change of flags or inlining.  -DD3IL_RARE_OUTLINE selects the out-of-line form.
#if defined(D3IL_RARE_OUTLINE)
#define D3IL_RARE D3IL_NOINLINE inline
#else
#define D3IL_RARE D3IL_HD
#endif

#if defined(D3IL_HOST_STATS)
#define D3IL_STAT(x) (x)
#define D3IL_DSTAT(i) ((void)0)
namespace d3il { struct Stats { long newton_calls, newton_iters, ls_iters, eig_calls, ik_calls, contact_calls; }; inline Stats g_stats = {0, 0, 0, 0, 0, 0}; }
#elif defined(D3IL_DEVICE_STATS) && defined(__HIPCC__)
// diagnostics build only (python -m d3il_amd.build --stats): counts lanes [2i] and waves [2i+1] entering rare paths
namespace d3il { __device__ unsigned long long g_dev_stats[32]; __device__ unsigned long long g_dev_wave[4096][10]; __device__ unsigned long long g_dev_cnt[4096][8]; }
#define D3IL_STAT(x) ((void)0)
#if !defined(__HIP_DEVICE_COMPILE__)
#define D3IL_DSTAT(i) ((void)0)
#else
#define D3IL_DSTAT(i) do { atomicAdd(&d3il::g_dev_stats[2 * (i)], 1ull); \
    if (__builtin_amdgcn_mbcnt_hi(__builtin_amdgcn_read_exec_hi(), __builtin_amdgcn_mbcnt_lo(__builtin_amdgcn_read_exec_lo(), 0u)) == 0) { atomicAdd(&d3il::g_dev_stats[2 * (i) + 1], 1ull); if (blockIdx.x < 4096) atomicAdd(&d3il::g_dev_wave[blockIdx.x][i], 1ull); } } while (0)
#endif
#else
#define D3IL_STAT(x) ((void)0)
#define D3IL_DSTAT(i) ((void)0)
#endif
// lane-level event counter of the diagnostics build (g_dev_stats[24 + slot]; 0: Rayleigh-quotient steps of the deflate path, 1: deflate solves that started from a warm vector)
#if defined(D3IL_DEVICE_STATS) && defined(__HIP_DEVICE_COMPILE__)
#define D3IL_DCOUNT(slot) atomicAdd(&d3il::g_dev_stats[24 + (slot)], 1ull)
#else
#define D3IL_DCOUNT(slot) ((void)0)
#endif

namespace d3il {

// Constant-block access.  Two instantiations of the math below exist on the device:
//  * C = PandaConsts (a constexpr object generated at build time, csrc/gen/<task>_consts.inc): every constant is a
//    compile-time literal, folded into instruction operands (zeros of the kinematic tree vanish, nothing is loaded);
//  * C = address_space(4) PandaConsts (runtime block in constant memory, scalar loads).  For that one the pointer is
//    re-materialised through an empty asm at phase boundaries so that the ~450 constant loads are not all hoisted to
//    the top of the sub-step (which exhausts the SGPR file and spills through VGPR lanes).
template <class C> D3IL_HD const C& refresh(const C& c) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (!std::is_same<C, PandaConsts>::value) {
    const C* p = &c;
    asm volatile("" : "+s"(p));
    return *p;
  } else {
    return c;
  }
#else
  return c;
#endif
}
#define D3IL_REFRESH(ref, name) const auto& name = refresh(ref)

// flag bits of EnvState::flags
enum : unsigned {
  F_MODE_MASK = 0x1FFu, F_L1 = 1u << 9, F_L2 = 1u << 10, F_L3 = 1u << 11, F_TERMINATED = 1u << 12,
  F_SUCCESS = 1u << 13, F_ROD_CONTACT = 1u << 14, F_IK_VALID = 1u << 15, F_SOLVER_FAIL = 1u << 16,
  F_MULTI_CONTACT = 1u << 17,
};

// number of f64 state fields per environment and their order in the SoA state buffer
// (shared with the oracle's orc_env_get_state): qpos[9] qvel[9] bias[7] tcp[3] ik_q[7] ik_qd[7]
constexpr int STATE_F64 = 42;

struct EnvState {
  double q[NDOF], v[NDOF], bias[NARM], tcp[3], ikq[NARM], ikqd[NARM];
  unsigned flags;
  int step;
};

// ------------------------------------------------------------------ tiny vector helpers
template <class TA, class TB> D3IL_HD void cross3(TA a, TB b, double* r) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
template <class TA, class TB> D3IL_HD double dot3(TA a, TB b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
// r = E v  (E row-major 3x3)
template <class TA, class TB> D3IL_HD void mulE(TA E, TB v, double* r) {
  double x = E[0] * v[0] + E[1] * v[1] + E[2] * v[2], y = E[3] * v[0] + E[4] * v[1] + E[5] * v[2], z = E[6] * v[0] + E[7] * v[1] + E[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
// r = E^T v
template <class TA, class TB> D3IL_HD void mulEt(TA E, TB v, double* r) {
  double x = E[0] * v[0] + E[3] * v[1] + E[6] * v[2], y = E[1] * v[0] + E[4] * v[1] + E[7] * v[2], z = E[2] * v[0] + E[5] * v[1] + E[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
// E = Q * Rz(angle): parent <- link
template <class TA> D3IL_HD void joint_rot(TA Q, double s, double c, double* E) {
#pragma unroll
  for (int r = 0; r < 3; r++) { E[3 * r] = c * Q[3 * r] + s * Q[3 * r + 1]; E[3 * r + 1] = c * Q[3 * r + 1] - s * Q[3 * r]; E[3 * r + 2] = Q[3 * r + 2]; }
}
// symmetric 3x3 (xx yy zz xy xz yz) times vector
template <class TA, class TB> D3IL_HD void sym3v(TA I, TB v, double* r) {
  double x = I[0] * v[0] + I[3] * v[1] + I[4] * v[2], y = I[3] * v[0] + I[1] * v[1] + I[5] * v[2], z = I[4] * v[0] + I[5] * v[1] + I[2] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
D3IL_HD double clampd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }
// 1/x.  Device: v_rcp_f64 (4.6e-8 relative) + two Newton steps = 1.1e-16 maximum relative error, the same bound as the
// IEEE division sequence (measured on MI355X over 1e6 arguments spanning 1e-9 .. 1e9) in 5 instructions instead of ~12.
D3IL_HD double rcpd(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
#else
  return 1.0 / x;
#endif
}
D3IL_HD double rsqrtd(double x) {   // 1 / sqrt(x), x > 0: v_rsq_f64 + two Newton steps on the device
#if defined(__HIP_DEVICE_COMPILE__)
  double y = __builtin_amdgcn_rsq(x);
  y = y * (1.5 - 0.5 * x * y * y);
  y = y * (1.5 - 0.5 * x * y * y);
  return y;
#else
  return 1.0 / sqrt(x);
#endif
}
D3IL_HD int tri(int r, int c) { return r * (r + 1) / 2 + c; }  // packed lower-triangular index, r >= c
// sin / cos of an angle that moved by a small dl (|dl| <= 0.01): angle addition with a Taylor series of the increment
// (truncation < 1e-25 relative); used to carry the joints' sin / cos from sub-step to sub-step instead of re-evaluating them
D3IL_HD void trig_advance(double dl, double& sn, double& cs) {
  const double d2 = dl * dl;
  const double sd = dl * (1.0 - d2 * (1.0 / 6.0) * (1.0 - d2 * (1.0 / 20.0) * (1.0 - d2 * (1.0 / 42.0))));
  const double cd = 1.0 - d2 * 0.5 * (1.0 - d2 * (1.0 / 12.0) * (1.0 - d2 * (1.0 / 30.0) * (1.0 - d2 * (1.0 / 56.0))));
  const double s2 = sn * cd + cs * sd, c2 = cs * cd - sn * sd;
  sn = s2; cs = c2;
}


// ------------------------------------------------------------------ controller kinematics (URDF chain, core/Model.py:37-66)
// pos/R of the grasp-target frame, world joint axes and origins
template <class C> D3IL_HD void ik_chain(const C& c0, const double* sq, const double* cq, double* p, double* R, double (*ax)[3], double (*og)[3]) {
  R[0] = 1; R[1] = 0; R[2] = 0; R[3] = 0; R[4] = 1; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1;
  p[0] = p[1] = p[2] = 0;
#pragma unroll
  for (int k = 0; k < NARM; k++) {
    D3IL_REFRESH(c0, c);
    double t[3]; mulE(R, c.Kx[k], t);
    p[0] += t[0]; p[1] += t[1]; p[2] += t[2];
    double Rn[9];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int cc = 0; cc < 3; cc++) Rn[3 * r + cc] = R[3 * r] * c.KR[k][cc] + R[3 * r + 1] * c.KR[k][3 + cc] + R[3 * r + 2] * c.KR[k][6 + cc];
    ax[k][0] = Rn[2]; ax[k][1] = Rn[5]; ax[k][2] = Rn[8];
    og[k][0] = p[0]; og[k][1] = p[1]; og[k][2] = p[2];
    joint_rot(Rn, sq[k], cq[k], R);
  }
  D3IL_REFRESH(c0, c);
  double t[3]; mulE(R, c.tool_x, t);
  p[0] += t[0]; p[1] += t[1]; p[2] += t[2];
  double Rn[9];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int cc = 0; cc < 3; cc++) Rn[3 * r + cc] = R[3 * r] * c.tool_R[cc] + R[3 * r + 1] * c.tool_R[3 + cc] + R[3 * r + 2] * c.tool_R[6 + cc];
#pragma unroll
  for (int i = 0; i < 9; i++) R[i] = Rn[i];
}
// Eigen::Quaternion(Matrix3) branch rule, what pinocchio.Quaternion(R) evaluates (Model.py:47-53) [ext]
D3IL_HD void mat2quat(const double* R, double* q) {
  double t = R[0] + R[4] + R[8];
  if (t > 0) {
    t = sqrt(t + 1.0); q[0] = 0.5 * t; t = 0.5 * rcpd(t);
    q[1] = (R[7] - R[5]) * t; q[2] = (R[2] - R[6]) * t; q[3] = (R[3] - R[1]) * t;
  } else if (R[0] >= R[4] && R[0] >= R[8]) {     // i = 0
    t = sqrt(R[0] - R[4] - R[8] + 1.0); q[1] = 0.5 * t; t = 0.5 * rcpd(t);
    q[0] = (R[7] - R[5]) * t; q[2] = (R[3] + R[1]) * t; q[3] = (R[6] + R[2]) * t;
  } else if (R[4] > R[0] && R[4] >= R[8]) {      // i = 1
    t = sqrt(R[4] - R[8] - R[0] + 1.0); q[2] = 0.5 * t; t = 0.5 * rcpd(t);
    q[0] = (R[2] - R[6]) * t; q[3] = (R[7] + R[5]) * t; q[1] = (R[1] + R[3]) * t;
  } else {                                        // i = 2
    t = sqrt(R[8] - R[0] - R[4] + 1.0); q[3] = 0.5 * t; t = 0.5 * rcpd(t);
    q[0] = (R[3] - R[1]) * t; q[1] = (R[2] + R[6]) * t; q[2] = (R[5] + R[7]) * t;
  }
}
D3IL_HD void quat_error(const double* c, const double* d, double* e) {  // utils/geometric_transformation.py:14-46
  e[0] = c[0] * d[1] - d[0] * c[1] - c[3] * d[2] + c[2] * d[3];
  e[1] = c[0] * d[2] - d[0] * c[2] + c[3] * d[1] - c[1] * d[3];
  e[2] = c[0] * d[3] - d[0] * c[3] - c[2] * d[1] + c[1] * d[2];
}

// x = (U clip(S) U^T)^-1 b for the SPD 6x6 A (packed lower, 21): the reference rebuilds the matrix from its
// SVD with clipped singular values and calls np.linalg.solve (IKControllers.py:230-266).  For an SPD matrix this is
// x = sum_i (v_i . b) / clip(l_i, lo, hi) v_i over its eigenpairs.  Three paths, all evaluating that same function:
//  1. no eigenvalue outside [lo, hi]  ->  x = A^-1 b (LDL^T).  The number of eigenvalues below lo is the number of
//     negative pivots of LDL^T(A - lo I) (Sylvester's law of inertia, exact); trace(A) < hi bounds the largest.
//  2. exactly one eigenvalue l1 < lo  ->  x = A^-1 b + (1/lo - 1/l1) (v1 . b) v1 with (l1, v1) from inverse
//     iteration followed by Rayleigh-quotient iteration; accepted only if the eigen-residual is at round-off level.
//  3. anything else (or path 2 not accepted)  ->  cyclic Jacobi eigen-decomposition.
D3IL_HD bool ldl6(const double* A, double shift, double* L, double* d, double* id, int* n_neg) {
  bool ok = true; int neg = 0;
#pragma unroll
  for (int j = 0; j < 6; j++) {
    double s = A[tri(j, j)] - shift;
#pragma unroll
    for (int k = 0; k < j; k++) s -= L[tri(j, k)] * L[tri(j, k)] * d[k];
    if (fabs(s) < 1e-30) { s = 1e-30; ok = false; }
    d[j] = s; neg += s < 0 ? 1 : 0;
    double inv = rcpd(s);
    id[j] = inv;
#pragma unroll
    for (int i = j + 1; i < 6; i++) {
      double t = A[tri(i, j)];
#pragma unroll
      for (int k = 0; k < j; k++) t -= L[tri(i, k)] * L[tri(j, k)] * d[k];
      L[tri(i, j)] = t * inv;
    }
  }
  *n_neg = neg;
  return ok;
}
D3IL_HD void ldl6_solve(const double* L, const double* id, const double* b, double* x) {
  double y[6];
#pragma unroll
  for (int i = 0; i < 6; i++) { double s = b[i];
#pragma unroll
    for (int k = 0; k < i; k++) s -= L[tri(i, k)] * y[k]; y[i] = s; }
#pragma unroll
  for (int i = 0; i < 6; i++) y[i] *= id[i];
#pragma unroll
  for (int i = 5; i >= 0; i--) { double s = y[i];
#pragma unroll
    for (int k = i + 1; k < 6; k++) s -= L[tri(k, i)] * x[k]; x[i] = s; }
}
D3IL_HD void symv6(const double* A, const double* x, double* y) {
#pragma unroll
  for (int i = 0; i < 6; i++) { double s = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) s += A[i >= k ? tri(i, k) : tri(k, i)] * x[k]; y[i] = s; }
}
D3IL_NOINLINE inline void jacobi_solve6(const double* A, const double* b, double minsv, double maxsv, double* x) {
  D3IL_STAT(g_stats.eig_calls++);
  D3IL_DSTAT(0);
  double M[6][6], V[6][6];
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) { M[i][j] = A[i >= j ? tri(i, j) : tri(j, i)]; V[i][j] = i == j ? 1.0 : 0.0; }
  for (int sweep = 0; sweep < 30; sweep++) {
    double off = 0, dg = 0;
    for (int i = 0; i < 6; i++) { dg += M[i][i] * M[i][i];
      for (int j = i + 1; j < 6; j++) off += M[i][j] * M[i][j]; }
    if (off <= 1e-34 * dg) break;
    for (int p = 0; p < 5; p++)
      for (int q = p + 1; q < 6; q++) {
        double apq = M[p][q];
        if (apq != 0.0) {
          double theta = (M[q][q] - M[p][p]) / (2 * apq);
          double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
          double cs = 1 / sqrt(t * t + 1), sn = t * cs;
          for (int k = 0; k < 6; k++) { double a = M[k][p], bb = M[k][q]; M[k][p] = cs * a - sn * bb; M[k][q] = sn * a + cs * bb; }
          for (int k = 0; k < 6; k++) { double a = M[p][k], bb = M[q][k]; M[p][k] = cs * a - sn * bb; M[q][k] = sn * a + cs * bb; }
          for (int k = 0; k < 6; k++) { double a = V[k][p], bb = V[k][q]; V[k][p] = cs * a - sn * bb; V[k][q] = sn * a + cs * bb; }
        }
      }
  }
  double y[6];
  for (int i = 0; i < 6; i++) { double s = 0;
    for (int a = 0; a < 6; a++) s += V[a][i] * b[a]; y[i] = s / clampd(M[i][i], minsv, maxsv); }
  for (int a = 0; a < 6; a++) { double s = 0;
    for (int i = 0; i < 6; i++) s += V[a][i] * y[i]; x[a] = s; }
}
// Path 2 of ik_solve6 (exactly one eigenvalue below lo): L, id = LDL^T(A), tr = trace(A).  Returns whether the eigenpair was accepted (x written).
D3IL_HD bool ik_deflate(const double* A, const double* b, const double* L, const double* id, double tr, double minsv, double* x, double* vwarm) {
  bool accepted = false;
  // smallest eigenpair: two inverse-iteration steps starting from b, then Rayleigh-quotient iteration
  D3IL_DSTAT(1);
  // start vector: the eigenvector accepted by the previous solve of this environment if there is one (the matrix
  // changes little between IK iterations), else two inverse-iteration steps starting from b
  double v[6], w[6], nr = 0;
  if (vwarm && vwarm[6] != 0.0) {
    D3IL_DCOUNT(1);
#pragma unroll
    for (int i = 0; i < 6; i++) v[i] = vwarm[i];
  } else {
    ldl6_solve(L, id, b, w);
    ldl6_solve(L, id, w, v);
#pragma unroll
    for (int i = 0; i < 6; i++) nr += v[i] * v[i];
    nr = 1.0 / sqrt(nr);
#pragma unroll
    for (int i = 0; i < 6; i++) v[i] *= nr;
  }
  double lam = 0, res = 0, vb = 0;
  for (int it = 0; it < 5; it++) {
    symv6(A, v, w);
    lam = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) lam += v[i] * w[i];
    res = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) res = fmax(res, fabs(w[i] - lam * v[i]));
    if (res <= 1e-14 * tr || it == 4) break;
    D3IL_DCOUNT(0);
    double L2[21], d2[6], id2[6], u[6];
    int n2;
    ldl6(A, lam, L2, d2, id2, &n2);
    ldl6_solve(L2, id2, v, u);
    nr = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) nr += u[i] * u[i];
    nr = 1.0 / sqrt(nr);
#pragma unroll
    for (int i = 0; i < 6; i++) v[i] = u[i] * nr;
  }
#pragma unroll
  for (int i = 0; i < 6; i++) vb += v[i] * b[i];
  if (res <= 1e-14 * tr && lam < minsv && lam > 0) {
    // x = A^-1 (b - (v.b) v) + (v.b)/lo v : the clipped direction is removed BEFORE the solve (no cancellation)
    double bp[6], xp[6], vx = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) bp[i] = b[i] - vb * v[i];
    ldl6_solve(L, id, bp, xp);
#pragma unroll
    for (int i = 0; i < 6; i++) vx += v[i] * xp[i];
    double g = vb / minsv - vx;
#pragma unroll
    for (int i = 0; i < 6; i++) x[i] = xp[i] + g * v[i];
    accepted = true;
    if (vwarm) {
#pragma unroll
      for (int i = 0; i < 6; i++) vwarm[i] = v[i];
      vwarm[6] = 1.0;
    }
  }
  return accepted;
}
template <bool FAST>
D3IL_HD void ik_solve6(const double* A, const double* b, double minsv, double maxsv, double* x, double* vwarm = nullptr) {
  bool need_eig = true;
  if (FAST) {
    double L[21], d[6], id[6];
    int neg = 0;
    bool okp = ldl6(A, minsv, L, d, id, &neg);
    double tr = A[tri(0, 0)] + A[tri(1, 1)] + A[tri(2, 2)] + A[tri(3, 3)] + A[tri(4, 4)] + A[tri(5, 5)];
    if (okp && tr < maxsv && neg <= 1) {
      int dummy;
      bool ok0 = ldl6(A, 0.0, L, d, id, &dummy);
      if (neg == 0) { ldl6_solve(L, id, b, x); need_eig = !ok0; }
      else if (ok0) {
        need_eig = !ik_deflate(A, b, L, id, tr, minsv, x, vwarm);
      }
    }
  }
  if (need_eig) {
    if (vwarm) vwarm[6] = 0.0;
    double Am[21], bm[6], xm[6];
#pragma unroll
    for (int i = 0; i < 21; i++) Am[i] = A[i];
#pragma unroll
    for (int i = 0; i < 6; i++) bm[i] = b[i];
    jacobi_solve6(Am, bm, minsv, maxsv, xm);
#pragma unroll
    for (int i = 0; i < 6; i++) x[i] = xm[i];
  }
}

// One call of CartPosQuatImpedenceController.getControl up to (and including) the set-point handed to the
// joint PD law: advances the virtual joint target ikq by ik_iters damped-least-squares iterations.
template <bool FAST, class C>
// trig (optional, 15 doubles): sin[7] cos[7] of ikq carried between calls, [14] != 0 when valid
D3IL_HD void ik_update(const C& c0, const double* des_pos, const double* des_quat_in, const double* cur_q,
                       unsigned& flags, double* ikq, double* ikqd, double* vwarm = nullptr, double* trig = nullptr) {
  double q[NARM], old_q[NARM];
  if (!(flags & F_IK_VALID)) {
#pragma unroll
    for (int k = 0; k < NARM; k++) ikq[k] = cur_q[k];
    flags |= F_IK_VALID;
    if (trig) trig[2 * NARM] = 0.0;
  }
#pragma unroll
  for (int k = 0; k < NARM; k++) { old_q[k] = ikq[k]; q[k] = ikq[k]; }
  double dq[4] = {des_quat_in[0], des_quat_in[1], des_quat_in[2], des_quat_in[3]};
  // sin/cos of the virtual joint angles: exact once per call, then advanced by the angle-addition formulas with a
  // short Taylor series of the (tiny) increment: |dq| <= ik_lr * 3 (norm clip) - truncation error < 1e-25 relative.
  double sq[NARM], cq[NARM];
  const bool small_step = c0.ik_lr * 3.0 <= 0.01;
  const bool carry = trig != nullptr && small_step;
  if (carry && trig[2 * NARM] != 0.0) {
#pragma unroll
    for (int k = 0; k < NARM; k++) { sq[k] = trig[k]; cq[k] = trig[NARM + k]; }
  } else {
#pragma unroll
    for (int k = 0; k < NARM; k++) sincos(q[k], &sq[k], &cq[k]);
  }
  const int n_it = c0.ik_iters;
#pragma clang loop unroll(disable)
  for (int it = 0; it < n_it; it++) {
    D3IL_REFRESH(c0, c);
    double pos[3], R[9], ax[NARM][3], og[NARM][3], cq4[4];
    ik_chain(c, sq, cq, pos, R, ax, og);
    mat2quat(R, cq4);
    double dm = 0, dp = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { dm += (cq4[k] - dq[k]) * (cq4[k] - dq[k]); dp += (cq4[k] + dq[k]) * (cq4[k] + dq[k]); }
    if (dm > dp) { dq[0] = -dq[0]; dq[1] = -dq[1]; dq[2] = -dq[2]; dq[3] = -dq[3]; }   // norm(a) > norm(b) <=> a.a > b.b
    double qe[3], target[6];
    quat_error(cq4, dq, qe);
#pragma unroll
    for (int k = 0; k < 3; k++) {
      target[k] = c.ik_ppos[k] * clampd(des_pos[k] - pos[k], -0.01, 0.01);
      target[3 + k] = c.ik_pquat[k] * clampd(qe[k], -0.1, 0.1);
    }
    double J[6][NARM];
#pragma unroll
    for (int k = 0; k < NARM; k++) {
      double d[3] = {pos[0] - og[k][0], pos[1] - og[k][1], pos[2] - og[k][2]}, cr[3];
      cross3(ax[k], d, cr);
      J[0][k] = cr[0]; J[1][k] = cr[1]; J[2][k] = cr[2]; J[3][k] = ax[k][0]; J[4][k] = ax[k][1]; J[5][k] = ax[k][2];
    }
    double A[21];
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
      for (int b = 0; b <= a; b++) {
        double s = a == b ? c.ik_Jreg : 0.0;
#pragma unroll
        for (int k = 0; k < NARM; k++) s += J[a][k] * J[b][k];
        A[tri(a, b)] = s;
      }
    double qn[NARM], rhs[6], x[6];
#pragma unroll
    for (int k = 0; k < NARM; k++) qn[k] = c.ik_pnull[k] * clampd(c.ik_rest[k] - q[k], -0.2, 0.2);
#pragma unroll
    for (int a = 0; a < 6; a++) { double s = target[a];
#pragma unroll
      for (int k = 0; k < NARM; k++) s -= J[a][k] * qn[k]; rhs[a] = s; }
    ik_solve6<FAST>(A, rhs, c.ik_minsv, c.ik_maxsv, x, vwarm);
    double qd[NARM], nrm = 0;
#pragma unroll
    for (int k = 0; k < NARM; k++) { double s = qn[k];
#pragma unroll
      for (int a = 0; a < 6; a++) s += J[a][k] * x[a]; qd[k] = s; nrm += s * s; }
    nrm = sqrt(nrm);
    if (nrm > 3) {
#pragma unroll
      for (int k = 0; k < NARM; k++) qd[k] = qd[k] * 3 / nrm;
    }
#pragma unroll
    for (int k = 0; k < NARM; k++) {
      double qn2 = clampd(q[k] + c.ik_lr * qd[k], c.q_min[k], c.q_max[k]);
      if (it + 1 < n_it || carry) {
        if (small_step) trig_advance(qn2 - q[k], sq[k], cq[k]);
        else sincos(qn2, &sq[k], &cq[k]);
      }
      q[k] = qn2;
    }
  }
  if (carry) {
#pragma unroll
    for (int k = 0; k < NARM; k++) { trig[k] = sq[k]; trig[NARM + k] = cq[k]; }
    trig[2 * NARM] = 1.0;
  }
#pragma unroll
  for (int k = 0; k < NARM; k++) { ikqd[k] = (q[k] - old_q[k]) / c0.timestep; ikq[k] = q[k]; }
}

// ------------------------------------------------------------------ dynamics (MJCF chain), link-frame recursions
struct DynOut {
  double M[45];        // packed lower, dof order: 7 arm, finger1, finger2
  double bias[NDOF];   // Coriolis + centrifugal + gravity (qfrc_bias)
  double R7[9], p7[3]; // world <- link 7
  double sn[NARM], cs[NARM];  // sin/cos of the arm joints (reused by the contact Jacobian)
};

// world pose of link 7 plus (optionally) world joint axes / origins for point Jacobians
template <class C> D3IL_HD void world_chain(const C& c0, const double* sn, const double* cs, double* R7, double* p7, double (*ax)[3], double (*og)[3]) {
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, p[3] = {0, 0, 0};
#pragma unroll
  for (int i = 0; i < NARM; i++) {
    D3IL_REFRESH(c0, c);
    double t[3]; mulE(R, c.P[i], t);
    p[0] += t[0]; p[1] += t[1]; p[2] += t[2];
    double E[9], Rn[9];
    joint_rot(c.Q[i], sn[i], cs[i], E);
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int cc = 0; cc < 3; cc++) Rn[3 * r + cc] = R[3 * r] * E[cc] + R[3 * r + 1] * E[3 + cc] + R[3 * r + 2] * E[6 + cc];
#pragma unroll
    for (int k = 0; k < 9; k++) R[k] = Rn[k];
    if (ax) { ax[i][0] = R[2]; ax[i][1] = R[5]; ax[i][2] = R[8]; og[i][0] = p[0]; og[i][1] = p[1]; og[i][2] = p[2]; }
  }
#pragma unroll
  for (int k = 0; k < 9; k++) R7[k] = R[k];
  p7[0] = p[0]; p7[1] = p[1]; p7[2] = p[2];
}

// trig (optional): sin[7] cos[7] of the arm joints carried by the caller (physics_substep keeps them current)
template <class C> D3IL_HD void dynamics(const C& c0, const double* q, const double* v, DynOut& o, const double* trig = nullptr) {
  double sn[NARM], cs[NARM];
#pragma unroll
  for (int i = 0; i < NARM; i++) {
    if (trig) { sn[i] = trig[i]; cs[i] = trig[NARM + i]; } else sincos(q[i], &sn[i], &cs[i]);
    o.sn[i] = sn[i]; o.cs[i] = cs[i];
  }
  world_chain(c0, sn, cs, o.R7, o.p7, nullptr, nullptr);

  // ---- RNEA forward pass (velocities, accelerations with qacc = 0, base acceleration = -gravity)
  double F[NARM][3], N[NARM][3];   // link force and moment about the link origin, link axes
  double w[3] = {0, 0, 0}, al[3] = {0, 0, 0}, a[3] = {-c0.gravity[0], -c0.gravity[1], -c0.gravity[2]};
  double w7[3], al7[3], a7[3];
#pragma unroll
  for (int i = 0; i < NARM; i++) {
    D3IL_REFRESH(c0, c);
    double E[9]; joint_rot(c.Q[i], sn[i], cs[i], E);
    double t1[3], t2[3], ap[3];
    cross3(al, c.P[i], t1); cross3(w, c.P[i], t2); cross3(w, t2, t2);
    ap[0] = a[0] + t1[0] + t2[0]; ap[1] = a[1] + t1[1] + t2[1]; ap[2] = a[2] + t1[2] + t2[2];
    double wn[3], aln[3];
    mulEt(E, w, wn); wn[2] += v[i];
    mulEt(E, al, aln); aln[0] += wn[1] * v[i]; aln[1] -= wn[0] * v[i];
    mulEt(E, ap, a);
    w[0] = wn[0]; w[1] = wn[1]; w[2] = wn[2]; al[0] = aln[0]; al[1] = aln[1]; al[2] = aln[2];
    // body wrench
    double ac[3], Iw[3], Ia[3];
    cross3(al, c.com[i], t1); cross3(w, c.com[i], t2); cross3(w, t2, t2);
    ac[0] = a[0] + t1[0] + t2[0]; ac[1] = a[1] + t1[1] + t2[1]; ac[2] = a[2] + t1[2] + t2[2];
    F[i][0] = c.mass[i] * ac[0]; F[i][1] = c.mass[i] * ac[1]; F[i][2] = c.mass[i] * ac[2];
    sym3v(c.Ic[i], w, Iw); sym3v(c.Ic[i], al, Ia);
    cross3(w, Iw, t1); cross3(c.com[i], F[i], t2);
    N[i][0] = Ia[0] + t1[0] + t2[0]; N[i][1] = Ia[1] + t1[1] + t2[1]; N[i][2] = Ia[2] + t1[2] + t2[2];
  }
  w7[0] = w[0]; w7[1] = w[1]; w7[2] = w[2]; al7[0] = al[0]; al7[1] = al[1]; al7[2] = al[2]; a7[0] = a[0]; a7[1] = a[1]; a7[2] = a[2];
  // fingers: prismatic leaves of link 7
  double rf[NFING][3];
#pragma unroll
  for (int k = 0; k < NFING; k++) {
    D3IL_REFRESH(c0, c);
    auto ax = c.f_axis[k];
    double qf = q[NARM + k], vf = v[NARM + k];
    rf[k][0] = c.f_com0[k][0] + ax[0] * qf; rf[k][1] = c.f_com0[k][1] + ax[1] * qf; rf[k][2] = c.f_com0[k][2] + ax[2] * qf;
    double t1[3], t2[3], t3[3], ac[3], Ff[3], Iw[3], Ia[3];
    cross3(al7, rf[k], t1); cross3(w7, rf[k], t2); cross3(w7, t2, t2); cross3(w7, ax, t3);
    ac[0] = a7[0] + t1[0] + t2[0] + 2 * vf * t3[0]; ac[1] = a7[1] + t1[1] + t2[1] + 2 * vf * t3[1]; ac[2] = a7[2] + t1[2] + t2[2] + 2 * vf * t3[2];
    Ff[0] = c.f_mass[k] * ac[0]; Ff[1] = c.f_mass[k] * ac[1]; Ff[2] = c.f_mass[k] * ac[2];
    o.bias[NARM + k] = dot3(Ff, ax);
    sym3v(c.f_Ic[k], w7, Iw); sym3v(c.f_Ic[k], al7, Ia);
    cross3(w7, Iw, t1); cross3(rf[k], Ff, t2);
    F[NARM - 1][0] += Ff[0]; F[NARM - 1][1] += Ff[1]; F[NARM - 1][2] += Ff[2];
    N[NARM - 1][0] += Ia[0] + t1[0] + t2[0]; N[NARM - 1][1] += Ia[1] + t1[1] + t2[1]; N[NARM - 1][2] += Ia[2] + t1[2] + t2[2];
  }
  // ---- RNEA backward pass
#pragma unroll
  for (int i = NARM - 1; i >= 0; i--) {
    o.bias[i] = N[i][2];
    if (i > 0) {
      D3IL_REFRESH(c0, c);
      double E[9]; joint_rot(c.Q[i], sn[i], cs[i], E);
      double f[3], n[3], t[3];
      mulE(E, F[i], f); mulE(E, N[i], n); cross3(c.P[i], f, t);
      F[i - 1][0] += f[0]; F[i - 1][1] += f[1]; F[i - 1][2] += f[2];
      N[i - 1][0] += n[0] + t[0]; N[i - 1][1] += n[1] + t[1]; N[i - 1][2] += n[2] + t[2];
    }
  }

  // ---- CRBA: composite inertia (mass, h = m c, Io about the link origin) from the tip down
  double cm = 0, ch[3] = {0, 0, 0}, cI[6] = {0, 0, 0, 0, 0, 0};
  // finger columns (constant in the link-7 frame: n = m (com0 x axis), f = m axis) and finger composites
  double colF[NFING][3], colN[NFING][3];
#pragma unroll
  for (int k = 0; k < NFING; k++) {
    D3IL_REFRESH(c0, c);
    auto ax = c.f_axis[k];
    double m = c.f_mass[k];
    colF[k][0] = m * ax[0]; colF[k][1] = m * ax[1]; colF[k][2] = m * ax[2];
    cross3(rf[k], colF[k], colN[k]);
    o.M[tri(NARM + k, NARM + k)] = m;
    double rr = dot3(rf[k], rf[k]);
    cm += m; ch[0] += m * rf[k][0]; ch[1] += m * rf[k][1]; ch[2] += m * rf[k][2];
    cI[0] += c.f_Ic[k][0] + m * (rr - rf[k][0] * rf[k][0]); cI[1] += c.f_Ic[k][1] + m * (rr - rf[k][1] * rf[k][1]); cI[2] += c.f_Ic[k][2] + m * (rr - rf[k][2] * rf[k][2]);
    cI[3] += c.f_Ic[k][3] - m * rf[k][0] * rf[k][1]; cI[4] += c.f_Ic[k][4] - m * rf[k][0] * rf[k][2]; cI[5] += c.f_Ic[k][5] - m * rf[k][1] * rf[k][2];
  }
  o.M[tri(NARM + 1, NARM)] = 0;
#pragma unroll
  for (int i = NARM - 1; i >= 0; i--) {
    D3IL_REFRESH(c0, c);
    // add link i's own inertia
    {
      auto cc = c.com[i]; double m = c.mass[i], rr = dot3(cc, cc);
      cm += m; ch[0] += m * cc[0]; ch[1] += m * cc[1]; ch[2] += m * cc[2];
      cI[0] += c.Ic[i][0] + m * (rr - cc[0] * cc[0]); cI[1] += c.Ic[i][1] + m * (rr - cc[1] * cc[1]); cI[2] += c.Ic[i][2] + m * (rr - cc[2] * cc[2]);
      cI[3] += c.Ic[i][3] - m * cc[0] * cc[1]; cI[4] += c.Ic[i][4] - m * cc[0] * cc[2]; cI[5] += c.Ic[i][5] - m * cc[1] * cc[2];
    }
    // column of joint i: unit acceleration about z through the origin
    double f[3] = {-ch[1], ch[0], 0}, n[3] = {cI[4], cI[5], cI[2]};
    o.M[tri(i, i)] = n[2];
    // finger columns enter the chain at link 7
    if (i == NARM - 1) {
#pragma unroll
      for (int k = 0; k < NFING; k++) o.M[tri(NARM + k, i)] = colN[k][2];
    }
    // propagate this joint's column up to the base
    {
      double ff[3] = {f[0], f[1], f[2]}, nn[3] = {n[0], n[1], n[2]};
#pragma unroll
      for (int j = i; j > 0; j--) {
        D3IL_REFRESH(c0, c);
        double E[9]; joint_rot(c.Q[j], sn[j], cs[j], E);
        double f2[3], n2[3], t[3];
        mulE(E, ff, f2); mulE(E, nn, n2); cross3(c.P[j], f2, t);
        ff[0] = f2[0]; ff[1] = f2[1]; ff[2] = f2[2]; nn[0] = n2[0] + t[0]; nn[1] = n2[1] + t[1]; nn[2] = n2[2] + t[2];
        o.M[tri(i, j - 1)] = nn[2];
      }
    }
    if (i == NARM - 1) {
#pragma unroll
      for (int k = 0; k < NFING; k++) {
        double ff[3] = {colF[k][0], colF[k][1], colF[k][2]}, nn[3] = {colN[k][0], colN[k][1], colN[k][2]};
#pragma unroll
        for (int j = NARM - 1; j > 0; j--) {
          D3IL_REFRESH(c0, c);
          double E[9]; joint_rot(c.Q[j], sn[j], cs[j], E);
          double f2[3], n2[3], t[3];
          mulE(E, ff, f2); mulE(E, nn, n2); cross3(c.P[j], f2, t);
          ff[0] = f2[0]; ff[1] = f2[1]; ff[2] = f2[2]; nn[0] = n2[0] + t[0]; nn[1] = n2[1] + t[1]; nn[2] = n2[2] + t[2];
          o.M[tri(NARM + k, j - 1)] = nn[2];
        }
      }
    }
    // move the composite to the parent frame
    if (i > 0) {
      D3IL_REFRESH(c0, c);
      double E[9]; joint_rot(c.Q[i], sn[i], cs[i], E);
      double h2[3]; mulE(E, ch, h2);
      // Io' = E Io E^T
      double I9[9] = {cI[0], cI[3], cI[4], cI[3], cI[1], cI[5], cI[4], cI[5], cI[2]}, T[9];
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int cc = 0; cc < 3; cc++) T[3 * r + cc] = E[3 * r] * I9[cc] + E[3 * r + 1] * I9[3 + cc] + E[3 * r + 2] * I9[6 + cc];
      double J0 = T[0] * E[0] + T[1] * E[1] + T[2] * E[2], J1 = T[3] * E[3] + T[4] * E[4] + T[5] * E[5], J2 = T[6] * E[6] + T[7] * E[7] + T[8] * E[8];
      double J3 = T[0] * E[3] + T[1] * E[4] + T[2] * E[5], J4 = T[0] * E[6] + T[1] * E[7] + T[2] * E[8], J5 = T[3] * E[6] + T[4] * E[7] + T[5] * E[8];
      auto P = c.P[i];
      double ph = dot3(P, h2), pp = dot3(P, P), s = 2 * ph + cm * pp;
      cI[0] = J0 + s - (2 * P[0] * h2[0] + cm * P[0] * P[0]); cI[1] = J1 + s - (2 * P[1] * h2[1] + cm * P[1] * P[1]); cI[2] = J2 + s - (2 * P[2] * h2[2] + cm * P[2] * P[2]);
      cI[3] = J3 - (P[0] * h2[1] + h2[0] * P[1] + cm * P[0] * P[1]); cI[4] = J4 - (P[0] * h2[2] + h2[0] * P[2] + cm * P[0] * P[2]); cI[5] = J5 - (P[1] * h2[2] + h2[1] * P[2] + cm * P[1] * P[2]);
      ch[0] = h2[0] + cm * P[0]; ch[1] = h2[1] + cm * P[1]; ch[2] = h2[2] + cm * P[2];
    }
  }
}

// packed LDL^T of a 9x9 SPD matrix: L unit lower (strict part stored), d pivots.  Returns false if a pivot <= 0.
D3IL_HD bool ldl9(const double* A, double* L, double* d, double* id) {
  bool ok = true;
#pragma unroll
  for (int j = 0; j < NDOF; j++) {
    double s = A[tri(j, j)];
#pragma unroll
    for (int k = 0; k < j; k++) s -= L[tri(j, k)] * L[tri(j, k)] * d[k];
    d[j] = s; ok = ok && (s > 0);
    double inv = rcpd(s);
    id[j] = inv;
#pragma unroll
    for (int i = j + 1; i < NDOF; i++) {
      double t = A[tri(i, j)];
#pragma unroll
      for (int k = 0; k < j; k++) t -= L[tri(i, k)] * L[tri(j, k)] * d[k];
      L[tri(i, j)] = t * inv;
    }
  }
  return ok;
}
D3IL_HD void ldl9_solve(const double* L, const double* id, double* x) {
#pragma unroll
  for (int i = 0; i < NDOF; i++) { double s = x[i];
#pragma unroll
    for (int k = 0; k < i; k++) s -= L[tri(i, k)] * x[k]; x[i] = s; }
#pragma unroll
  for (int i = 0; i < NDOF; i++) x[i] *= id[i];
#pragma unroll
  for (int i = NDOF - 1; i >= 0; i--) { double s = x[i];
#pragma unroll
    for (int k = i + 1; k < NDOF; k++) s -= L[tri(k, i)] * x[k]; x[i] = s; }
}
D3IL_HD void symv9(const double* A, const double* x, double* y) {
#pragma unroll
  for (int i = 0; i < NDOF; i++) { double s = 0;
#pragma unroll
    for (int k = 0; k < NDOF; k++) s += A[i >= k ? tri(i, k) : tri(k, i)] * x[k]; y[i] = s; }
}

// MuJoCo impedance d(r) from solimp (already clamped): engine_core_constraint getimpedance [ext]
template <class TA> D3IL_HD double impedance(TA si, double r) {
  if (si[0] == si[1] || si[2] <= 1e-15) return 0.5 * (si[0] + si[1]);
  double x = fabs(r) / si[2];
  if (x >= 1) return si[1];
  if (x <= 0) return si[0];
  double y;
  if (si[4] == 1) y = x;
  else if (x <= si[3]) y = (si[4] == 2 ? x * x : pow(x, si[4])) / (si[4] == 2 ? si[3] : pow(si[3], si[4] - 1));
  else { double u = 1 - x, m = 1 - si[3]; y = 1 - (si[4] == 2 ? u * u : pow(u, si[4])) / (si[4] == 2 ? m : pow(m, si[4] - 1)); }
  return si[0] + y * (si[1] - si[0]);
}

// rod <-> obstacle: capsule formula on the clamped closest points of the two axis segments
// (lateral regime; see DESIGN.md "Collision coverage").  Returns true if dist < margin.
template <class TA, class TB> D3IL_HD bool rod_obstacle(TA c1, TB u, double r1, double h1, const double* c2, const double* vv, double r2, double h2,
                          double margin, double* dist, double* nrm, double* pos) {
  double r[3] = {c1[0] - c2[0], c1[1] - c2[1], c1[2] - c2[2]};
  double b = dot3(u, vv), cc = dot3(u, r), f = dot3(vv, r), den = 1 - b * b, s, t;
  if (den > 1e-12) s = clampd((b * f - cc) / den, -h1, h1);
  else {
    double lo = fmax(-h1, -cc - h2), hi = fmin(h1, -cc + h2);
    s = lo <= hi ? 0.5 * (lo + hi) : (fabs(lo - h1) < fabs(hi + h1) ? h1 : -h1);
  }
  t = b * s + f;
  if (t < -h2) { t = -h2; s = clampd(b * t - cc, -h1, h1); }
  else if (t > h2) { t = h2; s = clampd(b * t - cc, -h1, h1); }
  double p1[3] = {c1[0] + s * u[0], c1[1] + s * u[1], c1[2] + s * u[2]};
  double d[3] = {c2[0] + t * vv[0] - p1[0], c2[1] + t * vv[1] - p1[1], c2[2] + t * vv[2] - p1[2]};
  double len = sqrt(dot3(d, d));
  *dist = len - r1 - r2;
  if (!(*dist < margin) || len < 1e-15) return false;
  nrm[0] = d[0] / len; nrm[1] = d[1] / len; nrm[2] = d[2] / len;
  double off = r1 + 0.5 * (*dist);
  pos[0] = p1[0] + nrm[0] * off; pos[1] = p1[1] + nrm[1] * off; pos[2] = p1[2] + nrm[2] * off;
  return true;
}

// data of the (single) rod contact handed to the solver
struct RodContact {
  bool active;
  double J[3][NARM];   // rows: normal, tangent1, tangent2 (elliptic cone, condim 3)
  double D[3], aref[3], mu, fric[2];
};

// Constraint solve: min_x 1/2 (x-a0)^T M (x-a0) + sum_i s_i(J_i x - aref_i) with joint-limit rows (unilateral
// quadratics) and at most one elliptic contact.  MuJoCo reaches the same (unique) optimum with its Newton solver.
// Returns qfrc_constraint; lim_* arrays are per dof (lim_sign = 0 when the row is absent).
// `warm` (10 doubles: previous solution + validity flag) carries the last optimum of this environment inside one env
// step; like MuJoCo's qacc_warmstart it only selects the starting point (whichever of warm / a0 has the lower cost),
// the optimum itself is unique.
D3IL_NOINLINE inline bool solve_constraints(const double* __restrict__ M_in, const double* __restrict__ a0_in, const double* __restrict__ fs_norm_ref,
                                            const double* __restrict__ lim_sign_in, const double* __restrict__ lim_D_in, const double* __restrict__ lim_aref_in,
                                            const RodContact& rc_in, double* __restrict__ fc_out, double* __restrict__ warm) {
  // operands arrive through private memory (out-of-line call): pull them into registers once
  double M[45], a0[NDOF], lim_sign[NDOF], lim_D[NDOF], lim_aref[NDOF];
#pragma unroll
  for (int i = 0; i < 45; i++) M[i] = M_in[i];
#pragma unroll
  for (int i = 0; i < NDOF; i++) { a0[i] = a0_in[i]; lim_sign[i] = lim_sign_in[i]; lim_D[i] = lim_D_in[i]; lim_aref[i] = lim_aref_in[i]; }
  RodContact rc;
  rc.active = rc_in.active;
  if (rc.active) {
#pragma unroll
    for (int r = 0; r < 3; r++) {
#pragma unroll
      for (int k = 0; k < NARM; k++) rc.J[r][k] = rc_in.J[r][k];
      rc.D[r] = rc_in.D[r]; rc.aref[r] = rc_in.aref[r];
    }
    rc.mu = rc_in.mu; rc.fric[0] = rc_in.fric[0]; rc.fric[1] = rc_in.fric[1];
  }
  double x[NDOF], Ma[NDOF], grad[NDOF], p[NDOF], fl[NDOF], hl[NDOF];
  double fcn[3] = {0, 0, 0}, Hc[6] = {0, 0, 0, 0, 0, 0};  // contact force and Hessian block (00 11 22 01 02 12)
#pragma unroll
  for (int i = 0; i < NDOF; i++) x[i] = a0[i];
  double gtol = 1e-11 * (1.0 + *fs_norm_ref);
  bool ok = true;
  D3IL_STAT(g_stats.newton_calls++);
  D3IL_STAT(g_stats.contact_calls += rc.active ? 1 : 0);
  D3IL_DSTAT(2);
  auto eval_contact = [&](const double* jar, double* force, double* H) {
    double mu = rc.mu, U0 = jar[0] * mu, U1 = jar[1] * rc.fric[0], U2 = jar[2] * rc.fric[1];
    double T = sqrt(U1 * U1 + U2 * U2), Nn = U0;
    force[0] = force[1] = force[2] = 0;
    if (H) { H[0] = H[1] = H[2] = H[3] = H[4] = H[5] = 0; }
    if (Nn >= mu * T || (T <= 0 && Nn >= 0)) return;
    if (mu * Nn + T <= 0 || (T <= 0 && Nn < 0)) {
      force[0] = -rc.D[0] * jar[0]; force[1] = -rc.D[1] * jar[1]; force[2] = -rc.D[2] * jar[2];
      if (H) { H[0] = rc.D[0]; H[1] = rc.D[1]; H[2] = rc.D[2]; }
      return;
    }
    double Dm = rc.D[0] / fmax(1e-15, mu * mu * (1 + mu * mu)), NmT = Nn - mu * T;
    double g0 = mu, g1 = -mu * rc.fric[0] * U1 / T, g2 = -mu * rc.fric[1] * U2 / T;
    force[0] = -Dm * NmT * g0; force[1] = -Dm * NmT * g1; force[2] = -Dm * NmT * g2;
    if (H) {
      double k = -mu * NmT, T3 = T * T * T;
      H[0] = Dm * g0 * g0; H[3] = Dm * g0 * g1; H[4] = Dm * g0 * g2;
      H[1] = Dm * (g1 * g1 + k * rc.fric[0] * rc.fric[0] * (1 / T - U1 * U1 / T3));
      H[2] = Dm * (g2 * g2 + k * rc.fric[1] * rc.fric[1] * (1 / T - U2 * U2 / T3));
      H[5] = Dm * (g1 * g2 + k * rc.fric[0] * rc.fric[1] * (-U1 * U2 / T3));
    }
  };
  if (warm[NDOF] != 0.0) {
    auto total_cost = [&](const double* y) {
      double dy[NDOF], My[NDOF], cst = 0;
#pragma unroll
      for (int i = 0; i < NDOF; i++) dy[i] = y[i] - a0[i];
      symv9(M, dy, My);
#pragma unroll
      for (int i = 0; i < NDOF; i++) {
        cst += 0.5 * dy[i] * My[i];
        double jar = lim_sign[i] * y[i] - lim_aref[i];
        if (lim_sign[i] != 0 && jar < 0) cst += 0.5 * lim_D[i] * jar * jar;
      }
      if (rc.active) {
        double jr[3];
#pragma unroll
        for (int r = 0; r < 3; r++) { double t = -rc.aref[r];
#pragma unroll
          for (int k = 0; k < NARM; k++) t += rc.J[r][k] * y[k]; jr[r] = t; }
        double mu = rc.mu, U1 = jr[1] * rc.fric[0], U2 = jr[2] * rc.fric[1], T = sqrt(U1 * U1 + U2 * U2), Nn = jr[0] * mu;
        if (Nn >= mu * T || (T <= 0 && Nn >= 0)) {}
        else if (mu * Nn + T <= 0 || (T <= 0 && Nn < 0)) cst += 0.5 * (rc.D[0] * jr[0] * jr[0] + rc.D[1] * jr[1] * jr[1] + rc.D[2] * jr[2] * jr[2]);
        else { double Dm = rc.D[0] / fmax(1e-15, mu * mu * (1 + mu * mu)), NmT = Nn - mu * T; cst += 0.5 * Dm * NmT * NmT; }
      }
      return cst;
    };
    double wl[NDOF];
#pragma unroll
    for (int i = 0; i < NDOF; i++) wl[i] = warm[i];
    if (total_cost(wl) < total_cost(a0)) {
#pragma unroll
      for (int i = 0; i < NDOF; i++) x[i] = wl[i];
    }
  }
  for (int it = 0; it < 12; it++) {
    // row residuals, forces, Hessian diagonal
    double jc[3] = {0, 0, 0};
#pragma unroll
    for (int i = 0; i < NDOF; i++) {
      double jar = lim_sign[i] * x[i] - lim_aref[i];
      bool act = lim_sign[i] != 0 && jar < 0;
      fl[i] = act ? -lim_D[i] * jar : 0.0; hl[i] = act ? lim_D[i] : 0.0;
    }
    if (rc.active) {
#pragma unroll
      for (int r = 0; r < 3; r++) { double s = -rc.aref[r];
#pragma unroll
        for (int k = 0; k < NARM; k++) s += rc.J[r][k] * x[k]; jc[r] = s; }
      eval_contact(jc, fcn, Hc);
    }
    double dx[NDOF];
#pragma unroll
    for (int i = 0; i < NDOF; i++) dx[i] = x[i] - a0[i];
    symv9(M, dx, Ma);
    double gn = 0;
#pragma unroll
    for (int i = 0; i < NDOF; i++) {
      double g = Ma[i] - lim_sign[i] * fl[i];
      if (rc.active && i < NARM) g -= rc.J[0][i] * fcn[0] + rc.J[1][i] * fcn[1] + rc.J[2][i] * fcn[2];
      grad[i] = g; gn += g * g;
    }
    if (sqrt(gn) <= gtol) break;
    D3IL_STAT(g_stats.newton_iters++);
    D3IL_DSTAT(3);
    // H = M + diag(hl) + Jc^T Hc Jc
    double H[45], L[45], d[NDOF], id[NDOF];
#pragma unroll
    for (int i = 0; i < 45; i++) H[i] = M[i];
#pragma unroll
    for (int i = 0; i < NDOF; i++) H[tri(i, i)] += hl[i];
    if (rc.active) {
      double T0[NARM], T1[NARM], T2[NARM];
#pragma unroll
      for (int k = 0; k < NARM; k++) {
        T0[k] = Hc[0] * rc.J[0][k] + Hc[3] * rc.J[1][k] + Hc[4] * rc.J[2][k];
        T1[k] = Hc[3] * rc.J[0][k] + Hc[1] * rc.J[1][k] + Hc[5] * rc.J[2][k];
        T2[k] = Hc[4] * rc.J[0][k] + Hc[5] * rc.J[1][k] + Hc[2] * rc.J[2][k];
      }
#pragma unroll
      for (int a = 0; a < NARM; a++)
#pragma unroll
        for (int b = 0; b <= a; b++) H[tri(a, b)] += rc.J[0][a] * T0[b] + rc.J[1][a] * T1[b] + rc.J[2][a] * T2[b];
    }
    if (!ldl9(H, L, d, id)) { ok = false; break; }
#pragma unroll
    for (int i = 0; i < NDOF; i++) p[i] = -grad[i];
    ldl9_solve(L, id, p);
    // exact line search on the convex 1-D restriction
    double Mp[NDOF], Jp[3] = {0, 0, 0};
    symv9(M, p, Mp);
    double pMp = 0, pMa = 0;
#pragma unroll
    for (int i = 0; i < NDOF; i++) { pMp += p[i] * Mp[i]; pMa += p[i] * Ma[i]; }
    if (rc.active) {
#pragma unroll
      for (int r = 0; r < 3; r++) { double s = 0;
#pragma unroll
        for (int k = 0; k < NARM; k++) s += rc.J[r][k] * p[k]; Jp[r] = s; }
    }
    double alpha = 0, lo = 0, hi = -1, best = 1, wprev = 1e300;
    for (int ls = 0; ls < 40; ls++) {
      D3IL_STAT(g_stats.ls_iters++);
      D3IL_DSTAT(6);
      double d1 = pMa + alpha * pMp, d2 = pMp;
#pragma unroll
      for (int i = 0; i < NDOF; i++) {
        double sp = lim_sign[i] * p[i];
        double jar = lim_sign[i] * x[i] - lim_aref[i] + alpha * sp;
        if (lim_sign[i] != 0 && jar < 0) { d1 += lim_D[i] * jar * sp; d2 += lim_D[i] * sp * sp; }
      }
      if (rc.active) {
        double jt[3] = {jc[0] + alpha * Jp[0], jc[1] + alpha * Jp[1], jc[2] + alpha * Jp[2]}, ft[3], Ht[6];
        eval_contact(jt, ft, Ht);
        d1 -= ft[0] * Jp[0] + ft[1] * Jp[1] + ft[2] * Jp[2];
        d2 += Ht[0] * Jp[0] * Jp[0] + Ht[1] * Jp[1] * Jp[1] + Ht[2] * Jp[2] * Jp[2] + 2 * (Ht[3] * Jp[0] * Jp[1] + Ht[4] * Jp[0] * Jp[2] + Ht[5] * Jp[1] * Jp[2]);
      }
      best = alpha;
      if (fabs(d1) <= 1e-14 * fmax(1.0, fabs(pMa))) break;
      if (d1 < 0) lo = alpha; else hi = alpha;
      double na = alpha - d1 / d2;
      if (hi >= 0) {   // bracketed: Newton on alpha, bisection whenever the bracket failed to halve (phi' can be sigmoid-like)
        double wbr = hi - lo;
        bool slow = wbr > 0.5 * wprev;
        wprev = wbr;
        if (slow || !(na > lo && na < hi)) na = 0.5 * (lo + hi);
      } else if (na <= lo) na = 2 * lo + 1;
      if (na == alpha) break;
      alpha = na;
    }
    double stepmax = 0, xmax = 0;
#pragma unroll
    for (int i = 0; i < NDOF; i++) { x[i] += best * p[i]; stepmax = fmax(stepmax, fabs(best * p[i])); xmax = fmax(xmax, fabs(x[i])); }
    if (stepmax <= 1e-13 * (1.0 + xmax)) break;   // converged to round-off: further iterations cannot move the iterate
  }
#pragma unroll
  for (int i = 0; i < NDOF; i++) warm[i] = x[i];
  warm[NDOF] = 1.0;
  // forces at the solution
#pragma unroll
  for (int i = 0; i < NDOF; i++) {
    double jar = lim_sign[i] * x[i] - lim_aref[i];
    fc_out[i] = (lim_sign[i] != 0 && jar < 0) ? -lim_D[i] * jar * lim_sign[i] : 0.0;
  }
  if (rc.active) {
    double jc[3];
#pragma unroll
    for (int r = 0; r < 3; r++) { double s = -rc.aref[r];
#pragma unroll
      for (int k = 0; k < NARM; k++) s += rc.J[r][k] * x[k]; jc[r] = s; }
    eval_contact(jc, fcn, nullptr);
#pragma unroll
    for (int k = 0; k < NARM; k++) fc_out[k] += rc.J[0][k] * fcn[0] + rc.J[1][k] * fcn[1] + rc.J[2][k] * fcn[2];
  }
  return ok;
}

#ifndef D3IL_LS_C2
#define D3IL_LS_C2 0.5     // curvature condition of the contact line search: |phi'(alpha)| <= c2 |phi'(0)|
#endif
// mju_makeFrame: tangents for a given normal [ext]
D3IL_HD void make_frame(const double* n, double* t1, double* t2) {
  double y[3] = {0, 0, 0};
  if (n[1] < 0.5 && n[1] > -0.5) y[1] = 1; else y[2] = 1;
  double d = dot3(n, y);
  y[0] -= d * n[0]; y[1] -= d * n[1]; y[2] -= d * n[2];
  double l = sqrt(dot3(y, y));
  t1[0] = y[0] / l; t1[1] = y[1] / l; t1[2] = y[2] / l;
  cross3(n, t1, t2);
}

// Contact sub-steps: one elliptic rod contact (3 rows) + the two finger-limit rows (D = 0 when absent) solved in the
// 5-dimensional constraint space.  At the optimum M (x - a0) = J^T f, so x = a0 + M^-1 J^T f and the row residuals are
// y = J x - aref = r + A f with A = J M^-1 J^T (5x5, SPD), r = J a0 - aref, f = -grad s(y).  Eliminating f, y minimises
//   Phi(y) = 1/2 (y - r)^T A^-1 (y - r) + s(y)      (strictly convex, same optimum as the 9-dof primal problem),
// which is minimised by Newton with an exact line search on 5x5 matrices.  Returns qfrc_constraint = J^T f.
// `warm` (6 doubles: f of the previous sub-step + validity) only selects the starting point.
D3IL_RARE bool solve_contact5(const double* __restrict__ L_in, const double* __restrict__ id_in, const double* __restrict__ a0_in,
                                         const RodContact& rc_in, const double* __restrict__ fsign, const double* __restrict__ fD,
                                         const double* __restrict__ faref, double gscale, double* __restrict__ fc_out, double* __restrict__ warm) {
  double L[45], id[NDOF], a0[NDOF];
#pragma unroll
  for (int i = 0; i < 45; i++) L[i] = L_in[i];
#pragma unroll
  for (int i = 0; i < NDOF; i++) { id[i] = id_in[i]; a0[i] = a0_in[i]; }
  double Jc[3][NARM], Dr[5], aref[5], s7 = fsign[0], s8 = fsign[1];
#pragma unroll
  for (int r = 0; r < 3; r++) {
#pragma unroll
    for (int k = 0; k < NARM; k++) Jc[r][k] = rc_in.J[r][k];
    Dr[r] = rc_in.D[r]; aref[r] = rc_in.aref[r];
  }
  Dr[3] = fD[0]; Dr[4] = fD[1]; aref[3] = faref[0]; aref[4] = faref[1];
  const double mu = rc_in.mu, fr0 = rc_in.fric[0], fr1 = rc_in.fric[1];
  const double Dm = rc_in.D[0] / fmax(1e-15, mu * mu * (1 + mu * mu));   // middle-zone stiffness of the cone
  // rows 3,4: finger limits, Jacobian s7 e_7 / s8 e_8 (unit vector with the limit's sign; +1 when the row is absent)
  const double j3 = s7 != 0 ? s7 : 1.0, j4 = s8 != 0 ? s8 : 1.0;
  // A = J M^-1 J^T and r = J a0 - aref, one column of M^-1 J^T at a time
  // A = J M^-1 J^T = (L^-1 J^T)^T D^-1 (L^-1 J^T): forward substitutions only
  double A[15], r[5], W[5][NDOF];
#pragma unroll
  for (int j = 0; j < 3; j++) {
#pragma unroll
    for (int i = 0; i < NDOF; i++) {
      double t = i < NARM ? Jc[j][i < NARM ? i : 0] : 0.0;
#pragma unroll
      for (int k = 0; k < i; k++) t -= L[tri(i, k)] * W[j][k];
      W[j][i] = t;
    }
  }
#pragma unroll
  for (int i = 0; i < NDOF; i++) { W[3][i] = 0; W[4][i] = 0; }
  W[3][7] = j3; W[3][8] = -L[tri(8, 7)] * j3; W[4][8] = j4;
#pragma unroll
  for (int j = 0; j < 5; j++)
#pragma unroll
    for (int i = j; i < 5; i++) {
      double t = 0;
#pragma unroll
      for (int k = (i >= 3 ? 7 : 0); k < NDOF; k++) t += W[i][k] * W[j][k] * id[k];
      A[tri(i, j)] = t;
    }
#pragma unroll
  for (int i = 0; i < 3; i++) { double t = -aref[i];
#pragma unroll
    for (int k = 0; k < NARM; k++) t += Jc[i][k] * a0[k]; r[i] = t; }
  r[3] = j3 * a0[7] - aref[3]; r[4] = j4 * a0[8] - aref[4];
  // Ainv (packed symmetric) from the LDL^T of A
  double LA[15], dA[5], idA[5], Ai[15];
  bool ok = true;
#pragma unroll
  for (int j = 0; j < 5; j++) {
    double t = A[tri(j, j)];
#pragma unroll
    for (int k = 0; k < j; k++) t -= LA[tri(j, k)] * LA[tri(j, k)] * dA[k];
    dA[j] = t; ok = ok && t > 0;
    double inv = rcpd(t);
    idA[j] = inv;
#pragma unroll
    for (int i = j + 1; i < 5; i++) {
      double v = A[tri(i, j)];
#pragma unroll
      for (int k = 0; k < j; k++) v -= LA[tri(i, k)] * LA[tri(j, k)] * dA[k];
      LA[tri(i, j)] = v * inv;
    }
  }
  auto solveA = [&](double* x) {
#pragma unroll
    for (int i = 0; i < 5; i++) { double t = x[i];
#pragma unroll
      for (int k = 0; k < i; k++) t -= LA[tri(i, k)] * x[k]; x[i] = t; }
#pragma unroll
    for (int i = 0; i < 5; i++) x[i] *= idA[i];
#pragma unroll
    for (int i = 4; i >= 0; i--) { double t = x[i];
#pragma unroll
      for (int k = i + 1; k < 5; k++) t -= LA[tri(k, i)] * x[k]; x[i] = t; }
  };
#pragma unroll
  for (int j = 0; j < 5; j++) {
    double e[5] = {0, 0, 0, 0, 0}; e[j] = 1; solveA(e);
#pragma unroll
    for (int i = j; i < 5; i++) Ai[tri(i, j)] = e[i];
  }
  // s(y), -grad (force) and Hessian blocks: cone on rows 0-2, unilateral quadratics on rows 3-4
  auto eval = [&](const double* y, double* f, double* Hc /*6*/, double* hl /*2*/, bool want_h) {
    double cst = 0;
    double U0 = y[0] * mu, U1 = y[1] * fr0, U2 = y[2] * fr1, T2 = U1 * U1 + U2 * U2, iT = T2 > 0 ? rsqrtd(T2) : 0.0, T = T2 * iT, Nn = U0;
    f[0] = f[1] = f[2] = 0;
    if (want_h) { Hc[0] = Hc[1] = Hc[2] = Hc[3] = Hc[4] = Hc[5] = 0; }
    if (Nn >= mu * T || (T <= 0 && Nn >= 0)) {}
    else if (mu * Nn + T <= 0 || (T <= 0 && Nn < 0)) {
      cst += 0.5 * (Dr[0] * y[0] * y[0] + Dr[1] * y[1] * y[1] + Dr[2] * y[2] * y[2]);
      f[0] = -Dr[0] * y[0]; f[1] = -Dr[1] * y[1]; f[2] = -Dr[2] * y[2];
      if (want_h) { Hc[0] = Dr[0]; Hc[1] = Dr[1]; Hc[2] = Dr[2]; }
    } else {
      double NmT = Nn - mu * T;
      double g0 = mu, g1 = -mu * fr0 * U1 * iT, g2 = -mu * fr1 * U2 * iT;
      cst += 0.5 * Dm * NmT * NmT;
      f[0] = -Dm * NmT * g0; f[1] = -Dm * NmT * g1; f[2] = -Dm * NmT * g2;
      if (want_h) {
        double k = -mu * NmT, iT3 = iT * iT * iT;
        Hc[0] = Dm * g0 * g0; Hc[3] = Dm * g0 * g1; Hc[4] = Dm * g0 * g2;
        Hc[1] = Dm * (g1 * g1 + k * fr0 * fr0 * (iT - U1 * U1 * iT3));
        Hc[2] = Dm * (g2 * g2 + k * fr1 * fr1 * (iT - U2 * U2 * iT3));
        Hc[5] = Dm * (g1 * g2 + k * fr0 * fr1 * (-U1 * U2 * iT3));
      }
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
      double yi = y[3 + i]; bool act = Dr[3 + i] != 0 && yi < 0;
      f[3 + i] = act ? -Dr[3 + i] * yi : 0.0;
      if (want_h) hl[i] = act ? Dr[3 + i] : 0.0;
      if (act) cst += 0.5 * Dr[3 + i] * yi * yi;
    }
    return cst;
  };
  auto symvAi = [&](const double* x, double* y) {
#pragma unroll
    for (int i = 0; i < 5; i++) { double t = 0;
#pragma unroll
      for (int k = 0; k < 5; k++) t += Ai[i >= k ? tri(i, k) : tri(k, i)] * x[k]; y[i] = t; }
  };
  auto phi = [&](const double* y) {
    double dy[5], w[5], f[5], t = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) dy[i] = y[i] - r[i];
    symvAi(dy, w);
#pragma unroll
    for (int i = 0; i < 5; i++) t += 0.5 * dy[i] * w[i];
    return t + eval(y, f, nullptr, nullptr, false);
  };
  D3IL_DSTAT(2);
  D3IL_STAT(g_stats.newton_calls++); D3IL_STAT(g_stats.contact_calls++);
  double y[5];
#pragma unroll
  for (int i = 0; i < 5; i++) y[i] = r[i];
  if (warm[5] != 0.0) {   // y_w = r + A f_w
    double yw[5];
#pragma unroll
    for (int i = 0; i < 5; i++) { double t = r[i];
#pragma unroll
      for (int k = 0; k < 5; k++) t += A[i >= k ? tri(i, k) : tri(k, i)] * warm[k]; yw[i] = t; }
    if (phi(yw) < phi(y)) {
#pragma unroll
      for (int i = 0; i < 5; i++) y[i] = yw[i];
    }
  }
  double f[5];
  const double gtol = 1e-10 * gscale;
  for (int it = 0; it < 12; it++) {
    double Hc[6], hl[2], dy[5], g[5];
    eval(y, f, Hc, hl, true);
#pragma unroll
    for (int i = 0; i < 5; i++) dy[i] = y[i] - r[i];
    symvAi(dy, g);
    double gn = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) { g[i] -= f[i]; gn += g[i] * g[i]; }
    // in force units: A g has the scale of an acceleration residual; compare the constraint-space gradient through A
    double Ag[5], an = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) { double t = 0;
#pragma unroll
      for (int k = 0; k < 5; k++) t += A[i >= k ? tri(i, k) : tri(k, i)] * g[k]; Ag[i] = t; an += t * t; }
    if (sqrt(an) <= gtol) break;
    D3IL_DSTAT(3); D3IL_STAT(g_stats.newton_iters++);
    // H = Ainv + Hs ; p = -H^-1 g   (5x5 LDL^T)
    double H[15], LH[15], dH[5], idH[5], p[5];
#pragma unroll
    for (int i = 0; i < 15; i++) H[i] = Ai[i];
    H[tri(0, 0)] += Hc[0]; H[tri(1, 1)] += Hc[1]; H[tri(2, 2)] += Hc[2]; H[tri(1, 0)] += Hc[3]; H[tri(2, 0)] += Hc[4]; H[tri(2, 1)] += Hc[5];
    H[tri(3, 3)] += hl[0]; H[tri(4, 4)] += hl[1];
#pragma unroll
    for (int j = 0; j < 5; j++) {
      double t = H[tri(j, j)];
#pragma unroll
      for (int k = 0; k < j; k++) t -= LH[tri(j, k)] * LH[tri(j, k)] * dH[k];
      dH[j] = t; ok = ok && t > 0;
      double inv = rcpd(t);
      idH[j] = inv;
#pragma unroll
      for (int i = j + 1; i < 5; i++) {
        double v = H[tri(i, j)];
#pragma unroll
        for (int k = 0; k < j; k++) v -= LH[tri(i, k)] * LH[tri(j, k)] * dH[k];
        LH[tri(i, j)] = v * inv;
      }
    }
#pragma unroll
    for (int i = 0; i < 5; i++) { double t = -g[i];
#pragma unroll
      for (int k = 0; k < i; k++) t -= LH[tri(i, k)] * p[k]; p[i] = t; }
#pragma unroll
    for (int i = 0; i < 5; i++) p[i] *= idH[i];
#pragma unroll
    for (int i = 4; i >= 0; i--) { double t = p[i];
#pragma unroll
      for (int k = i + 1; k < 5; k++) t -= LH[tri(k, i)] * p[k]; p[i] = t; }
    // exact line search
    double Aip[5], pAp = 0, pAd = 0;
    symvAi(p, Aip);
#pragma unroll
    for (int i = 0; i < 5; i++) { pAp += p[i] * Aip[i]; pAd += Aip[i] * dy[i]; }
    double gp0 = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) gp0 += g[i] * p[i];
    double alpha = 1, lo = 0, hi = -1, best = 1, wprev = 1e300;   // phi'(0) = g.p < 0 is known: start at the full Newton step
    for (int ls = 0; ls < 40; ls++) {
      D3IL_DSTAT(6); D3IL_STAT(g_stats.ls_iters++);
      double yt[5], ft[5], Ht[6], ht[2];
#pragma unroll
      for (int i = 0; i < 5; i++) yt[i] = y[i] + alpha * p[i];
      eval(yt, ft, Ht, ht, true);
      double d1 = pAd + alpha * pAp, d2 = pAp;
#pragma unroll
      for (int i = 0; i < 5; i++) d1 -= ft[i] * p[i];
      d2 += Ht[0] * p[0] * p[0] + Ht[1] * p[1] * p[1] + Ht[2] * p[2] * p[2] + 2 * (Ht[3] * p[0] * p[1] + Ht[4] * p[0] * p[2] + Ht[5] * p[1] * p[2]) + ht[0] * p[3] * p[3] + ht[1] * p[4] * p[4];
      best = alpha;
      // inexact line search: with the exact Hessian the full step passes this test near the optimum, so the outer
      // iteration keeps its quadratic rate; far from it a 1e-3 reduction of the directional derivative is plenty
      // the full Newton step is taken when phi is still descending there or just past its minimum (curvature condition)
      if (ls == 0 && d1 <= 0.1 * fabs(gp0)) break;
      if (fabs(d1) <= D3IL_LS_C2 * fabs(gp0)) break;
      if (d1 < 0) lo = alpha; else hi = alpha;
      double na = alpha - d1 * rcpd(d2);
      if (hi >= 0) {   // bracketed: Newton on alpha, bisection whenever the bracket failed to halve (phi' can be sigmoid-like)
        double wbr = hi - lo;
        bool slow = wbr > 0.5 * wprev;
        wprev = wbr;
        if (slow || !(na > lo && na < hi)) na = 0.5 * (lo + hi);
      } else if (na <= lo) na = 2 * lo + 1;
      if (na == alpha) break;
      alpha = na;
    }
    double stepmax = 0, ymax = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) { y[i] += best * p[i]; stepmax = fmax(stepmax, fabs(best * p[i])); ymax = fmax(ymax, fabs(y[i])); }
    // a full step below 1e-6 (relative) leaves an error of the order of its square: the confirming iteration is skipped
    if (stepmax <= 1e-13 * (1.0 + ymax) || (best == 1.0 && stepmax <= 1e-6 * (1.0 + ymax))) break;
  }
  eval(y, f, nullptr, nullptr, false);
#pragma unroll
  for (int i = 0; i < 5; i++) warm[i] = f[i];
  warm[5] = 1.0;
#pragma unroll
  for (int k = 0; k < NARM; k++) fc_out[k] = Jc[0][k] * f[0] + Jc[1][k] * f[1] + Jc[2][k] * f[2];
  fc_out[7] = j3 * f[3]; fc_out[8] = j4 * f[4];
  return ok;
}

// Rod contact set-up (rare path): Jacobian rows, regularisation and reference acceleration of the deepest contact.
template <class C>
D3IL_RARE void make_rod_contact(const C& c, const double* sn, const double* cs, const double* v, int bo, double bd, const double* bn, const double* bp, RodContact* rcp) {
  RodContact& rc = *rcp;
  rc.active = true;
  D3IL_DSTAT(5);
  double R7[9], p7[3], ax[NARM][3], og[NARM][3], t1[3], t2[3];
  world_chain(c, sn, cs, R7, p7, ax, og);
  make_frame(bn, t1, t2);
  double vel[3] = {0, 0, 0};
  for (int k = 0; k < NARM; k++) {
    double dd[3] = {bp[0] - og[k][0], bp[1] - og[k][1], bp[2] - og[k][2]}, col[3];
    cross3(ax[k], dd, col);
    rc.J[0][k] = dot3(bn, col); rc.J[1][k] = dot3(t1, col); rc.J[2][k] = dot3(t2, col);
    vel[0] += rc.J[0][k] * v[k]; vel[1] += rc.J[1][k] * v[k]; vel[2] += rc.J[2][k] * v[k];
  }
  double imp = impedance(c.ct_solimp[bo], bd - c.ct_margin[bo]);
  double Rn = fmax(1e-15, (1 - imp) / imp * c.rod_invweight0);
  double Rt = Rn / fmax(1e-15, c.impratio);
  double f0 = c.ct_fric[bo][0];
  rc.mu = f0 * sqrt(Rt / Rn); rc.fric[0] = f0; rc.fric[1] = f0;
  rc.D[0] = 1 / Rn; rc.D[1] = 1 / Rt; rc.D[2] = 1 / Rt;
  rc.aref[0] = -c.ct_B[bo] * vel[0] - c.ct_K[bo] * imp * (bd - c.ct_margin[bo]);
  rc.aref[1] = -c.ct_B[bo] * vel[1]; rc.aref[2] = -c.ct_B[bo] * vel[2];
}

// Where the two rare constraint paths (rod contact; arm joint-limit rows) run.  RareInline: in this function (every caller but the split
// Avoiding kernel).  A type with `remote == true` (RareXch, rollout.hip): the operands of the lanes that need a rare path go through an
// exchange area to ANOTHER WAVE of the workgroup, which runs rare_serve() below and hands the constraint force back - the hot path then
// carries none of the solvers' register pressure (192 spilled registers and 68 scratch operations per sub-step
// in the physics wave's main block with the solvers inlined, none without them).
struct RareInline { static constexpr bool remote = false; };
constexpr int RX_M = 0, RX_FS = 45, RX_Q = 54, RX_V = 63, RX_SN = 72, RX_MD = 86, RX_BO = 87, RX_BD = 88, RX_BN = 89, RX_BP = 92, RX_FL = 95 /* fsign[2] fD[2] faref[2] */,
              RX_ARM = 101, RX_L = 102 /* LDL^T factors of M: L[45], 1 / d[9] */, RX_ID = 147, RX_ROWS = 156, RX_FC = 0 /* reply: fc[9], fail */;

// The two rare constraint paths of physics_substep: a rod contact (+ the finger-limit rows) through the 5-dimensional constraint-space
// Newton, or - with an arm joint inside a limit margin - all limit rows + the rod contact through the 9-dof primal Newton solver.
// Operands: M / its LDL^T factors of this sub-step, fs = smooth force, the state, sin / cos of the arm joints, the deepest rod contact
// (bo < 0: none), the finger rows.  Writes qfrc_constraint; false = a solver gave up.
template <class C>
D3IL_HD bool rare_constraints(const C& c0, const double* M, const double* L, const double* id, const double* fs, const double* q, const double* v, const double* sn, const double* cs,
                              int bo, double bd, const double* bn, const double* bp, const double* fsign, const double* fD, const double* faref, bool arm_rows, double* fc, double* warm) {
  D3IL_REFRESH(c0, c);
  bool ok = true;
  if (!arm_rows) {
    double Lm[45], dm[NDOF], a0[NDOF], fcm[NDOF], vm[NDOF], fn = 0;
    for (int i = 0; i < 45; i++) Lm[i] = L[i];
    for (int k = 0; k < NDOF; k++) { dm[k] = id[k]; a0[k] = fs[k]; fn += fs[k] * fs[k]; vm[k] = v[k]; fcm[k] = 0; }
    ldl9_solve(L, id, a0);
    // gradient scale in acceleration units: |fs| / (mean diagonal of M)
    double md = 0;
    for (int k = 0; k < NDOF; k++) md += M[tri(k, k)];
    double gscale = (1.0 + sqrt(fn)) / (md / NDOF);
    RodContact rc; rc.active = false;
    make_rod_contact(c0, sn, cs, vm, bo, bd, bn, bp, &rc);
    ok = solve_contact5(Lm, dm, a0, rc, fsign, fD, faref, gscale, fcm, warm);
    for (int k = 0; k < NDOF; k++) fc[k] = fcm[k];
  } else {
    double Mm[45], a0[NDOF], lim_sign[NDOF], lim_D[NDOF], lim_aref[NDOF], fcm[NDOF], fn = 0, vm[NDOF];
    for (int i = 0; i < 45; i++) Mm[i] = M[i];
    for (int k = 0; k < NDOF; k++) { a0[k] = fs[k]; fn += fs[k] * fs[k]; vm[k] = v[k]; }
    fn = sqrt(fn);
    ldl9_solve(L, id, a0);
    for (int k = 0; k < NDOF; k++) {
      double dlo = q[k] - c.jnt_range[k][0], dhi = c.jnt_range[k][1] - q[k];
      double sign = 0, dist = 0;
      if (dlo < c.lim_margin[k]) { sign = 1; dist = dlo; }
      else if (dhi < c.lim_margin[k]) { sign = -1; dist = dhi; }
      lim_sign[k] = sign; lim_D[k] = 0; lim_aref[k] = 0;
      if (sign != 0) {
        double imp = impedance(c.lim_solimp[k], dist - c.lim_margin[k]);
        lim_D[k] = 1 / fmax(1e-15, (1 - imp) / imp * c.dof_invweight0[k]);
        lim_aref[k] = -c.lim_B[k] * (sign * v[k]) - c.lim_K[k] * imp * (dist - c.lim_margin[k]);
      }
    }
    RodContact rc; rc.active = false;
    if (bo >= 0) make_rod_contact(c0, sn, cs, vm, bo, bd, bn, bp, &rc);
    double warm9[NDOF + 1]; warm9[NDOF] = 0.0;
    ok = solve_constraints(Mm, a0, &fn, lim_sign, lim_D, lim_aref, rc, fcm, warm9);
    for (int k = 0; k < NDOF; k++) fc[k] = fcm[k];
  }
  return ok;
}

// The serving side of a remote rare path (one lane = one environment of the requesting wave, same lane index): operands (incl. the LDL^T
// factors of M the physics wave already has) from the exchange area, the reply (qfrc_constraint[9], failure flag) into rows RX_FC...
template <class C, class R> D3IL_HD void rare_serve(const C& c0, R* rare, double* warm) {
  double M[45], L[45], id[NDOF], fs[NDOF], q[NDOF], v[NDOF], sn[NARM], cs[NARM], bn[3], bp[3], fsign[NFING], fD[NFING], faref[NFING], fc[NDOF];
#pragma unroll
  for (int i = 0; i < 45; i++) { M[i] = rare->get(RX_M + i); L[i] = rare->get(RX_L + i); }
#pragma unroll
  for (int k = 0; k < NDOF; k++) id[k] = rare->get(RX_ID + k);
#pragma unroll
  for (int k = 0; k < NDOF; k++) { fs[k] = rare->get(RX_FS + k); q[k] = rare->get(RX_Q + k); v[k] = rare->get(RX_V + k); fc[k] = 0; }
#pragma unroll
  for (int k = 0; k < NARM; k++) { sn[k] = rare->get(RX_SN + k); cs[k] = rare->get(RX_SN + NARM + k); }
  const int bo = (int)rare->get(RX_BO);
  const double bd = rare->get(RX_BD);
#pragma unroll
  for (int k = 0; k < 3; k++) { bn[k] = rare->get(RX_BN + k); bp[k] = rare->get(RX_BP + k); }
#pragma unroll
  for (int k = 0; k < NFING; k++) { fsign[k] = rare->get(RX_FL + k); fD[k] = rare->get(RX_FL + 2 + k); faref[k] = rare->get(RX_FL + 4 + k); }
  const bool arm_rows = rare->get(RX_ARM) != 0.0;
  const bool ok = rare_constraints(c0, M, L, id, fs, q, v, sn, cs, bo, bd, bn, bp, fsign, fD, faref, arm_rows, fc, warm);      // a failed factorisation is flagged by the physics wave
#pragma unroll
  for (int k = 0; k < NDOF; k++) rare->put(RX_FC + k, fc[k]);
  rare->put(RX_FC + NDOF, ok ? 0.0 : 1.0);
}

// One mj_step (forward dynamics with the ctrl computed by the caller + semi-implicit Euler with implicit joint
// damping) followed by the state read-back.  `tau` = controller torque WITHOUT gravity compensation for the arm,
// `ffing` = raw finger command.  Updates q, v, bias (qfrc_bias of THIS forward pass), tcp (pre-integration pose).
//
// Constraint handling.  The soft-constraint problem min 1/2 (a-a0)'M(a-a0) + sum_i s_i(J_i a - aref_i) is strictly
// convex, so any exact method reproduces MuJoCo's Newton optimum.  Hot path: the only rows are finger joint limits
// (the fingers rest on their 0.04 m stop for most of an episode).  Their Jacobians are unit vectors on the last two
// dofs, so the optimum follows from the 2x2 block W = (M^-1)_FF, read off the LDL^T factors, by checking the four
// active sets - no iteration.  The same factors, with the last two pivots updated for the implicit damping term h B,
// give the integration solve: one 9x9 factorisation per sub-step.  Arm limit rows or a rod contact (rare) take the
// general out-of-line Newton path.
template <class C, class R = RareInline> D3IL_HD void physics_substep(const C& c0, EnvState& st, const double* tau, const double* ffing, double* warm, double* trig = nullptr, R* rare = nullptr) {
  D3IL_DSTAT(7);
  DynOut dyn;
  dynamics(c0, st.q, st.v, dyn, trig);
  D3IL_REFRESH(c0, c);
  // actuation: ctrl = tau + (stale) qfrc_bias for the arm, raw for fingers; motors clamp to forcerange
  double fs[NDOF];
#pragma unroll
  for (int k = 0; k < NARM; k++) fs[k] = clampd(tau[k] + st.bias[k], c.force_lo[k], c.force_hi[k]) - dyn.bias[k];
#pragma unroll
  for (int k = 0; k < NFING; k++) fs[NARM + k] = clampd(ffing[k], c.force_lo[NARM + k], c.force_hi[NARM + k]) - dyn.bias[NARM + k] - c.f_damping[k] * st.v[NARM + k];
  // read-back quantities of this forward pass (one sub-step stale w.r.t. the integrated state)
#pragma unroll
  for (int k = 0; k < NARM; k++) st.bias[k] = dyn.bias[k];
  {
    double t[3]; mulE(dyn.R7, c.tcp7, t);
    st.tcp[0] = dyn.p7[0] + t[0]; st.tcp[1] = dyn.p7[1] + t[1]; st.tcp[2] = dyn.p7[2] + t[2];
  }
  // ---- which constraints exist?
  bool arm_rows = false;
#pragma unroll
  for (int k = 0; k < NARM; k++) arm_rows = arm_rows || (st.q[k] - c.jnt_range[k][0] < c.lim_margin[k]) || (c.jnt_range[k][1] - st.q[k] < c.lim_margin[k]);
  double fsign[NFING], fD[NFING], faref[NFING];
#pragma unroll
  for (int k = 0; k < NFING; k++) {
    int j = NARM + k;
    double dlo = st.q[j] - c.jnt_range[j][0], dhi = c.jnt_range[j][1] - st.q[j];
    double sign = 0, dist = 0;
    if (dlo < c.lim_margin[j]) { sign = 1; dist = dlo; }
    else if (dhi < c.lim_margin[j]) { sign = -1; dist = dhi; }
    fsign[k] = sign; fD[k] = 0; faref[k] = 0;
    if (sign != 0) {
      double imp = impedance(c.lim_solimp[j], dist - c.lim_margin[j]);
      fD[k] = 1 / fmax(1e-15, (1 - imp) / imp * c.dof_invweight0[j]);
      faref[k] = -c.lim_B[j] * (sign * st.v[j]) - c.lim_K[j] * imp * (dist - c.lim_margin[j]);
    }
  }
  // rod <-> obstacle: deepest penetrating pair (geom1 = obstacle, lower geom id; normal points obstacle -> rod)
  double bd = 1e300, bn[3] = {0, 0, 0}, bp[3] = {0, 0, 0}; int bo = -1, ncon = 0;
  st.flags &= ~F_ROD_CONTACT;
  if (c0.n_obst > 0) {
    double rcw[3], ruw[3];
    mulE(dyn.R7, c.rod_c7, rcw); rcw[0] += dyn.p7[0]; rcw[1] += dyn.p7[1]; rcw[2] += dyn.p7[2];
    mulE(dyn.R7, c.rod_u7, ruw);
    for (int o = 0; o < c0.n_obst; o++) {
      double dist, nrm[3], pos[3];
      if (rod_obstacle(c.ob_c[o], c.ob_u[o], c.ob_r[o], c.ob_h[o], rcw, ruw, c.rod_r, c.rod_h, c.ct_margin[o], &dist, nrm, pos)) {
        ncon++;
        if (dist < bd) { bd = dist; bo = o; bn[0] = nrm[0]; bn[1] = nrm[1]; bn[2] = nrm[2]; bp[0] = pos[0]; bp[1] = pos[1]; bp[2] = pos[2]; }
      }
    }
    if (ncon > 1) st.flags |= F_MULTI_CONTACT;
    if (bo >= 0) st.flags |= F_ROD_CONTACT;
  }
  // ---- factorise M once
  double L[45], d[NDOF], id[NDOF];
  if (!ldl9(dyn.M, L, d, id)) st.flags |= F_SOLVER_FAIL;
  double fc[NDOF];
#pragma unroll
  for (int k = 0; k < NDOF; k++) fc[k] = 0;
  bool delegated = false;
  if constexpr (R::remote) {
    delegated = bo >= 0 || arm_rows;
    if (delegated) {
#pragma unroll
      for (int i = 0; i < 45; i++) rare->put(RX_M + i, dyn.M[i]);
#pragma unroll
      for (int k = 0; k < NDOF; k++) { rare->put(RX_FS + k, fs[k]); rare->put(RX_Q + k, st.q[k]); rare->put(RX_V + k, st.v[k]); }
#pragma unroll
      for (int k = 0; k < NARM; k++) { rare->put(RX_SN + k, dyn.sn[k]); rare->put(RX_SN + NARM + k, dyn.cs[k]); }
      rare->put(RX_BO, (double)bo); rare->put(RX_BD, bd);
#pragma unroll
      for (int k = 0; k < 3; k++) { rare->put(RX_BN + k, bn[k]); rare->put(RX_BP + k, bp[k]); }
#pragma unroll
      for (int k = 0; k < NFING; k++) { rare->put(RX_FL + k, fsign[k]); rare->put(RX_FL + 2 + k, fD[k]); rare->put(RX_FL + 4 + k, faref[k]); }
      rare->put(RX_ARM, arm_rows ? 1.0 : 0.0);
#pragma unroll
      for (int i = 0; i < 45; i++) rare->put(RX_L + i, L[i]);
#pragma unroll
      for (int k = 0; k < NDOF; k++) rare->put(RX_ID + k, id[k]);
    }
    if (rare->post(delegated) && delegated) {
#pragma unroll
      for (int k = 0; k < NDOF; k++) fc[k] = rare->get(RX_FC + k);
      if (rare->get(RX_FC + NDOF) != 0.0) st.flags |= F_SOLVER_FAIL;
    }
  }
  if (delegated) {
  } else if (!R::remote && (bo >= 0 || arm_rows)) {
    if (!rare_constraints(c0, dyn.M, L, id, fs, st.q, st.v, dyn.sn, dyn.cs, bo, bd, bn, bp, fsign, fD, faref, arm_rows, fc, warm)) st.flags |= F_SOLVER_FAIL;
  } else if (fsign[0] != 0 || fsign[1] != 0) {
    // finger-limit rows only: exact active-set solution on the 2x2 block W = (M^-1)_FF
    D3IL_STAT(g_stats.newton_calls++);
    D3IL_DSTAT(4);
    double l87 = L[tri(8, 7)];
    double W00 = id[7] + l87 * l87 * id[8], W01 = -l87 * id[8], W11 = id[8];
    // a0_F = (M^-1 fs)_F : full solve needed (a0 depends on all of fs)
    double a0[NDOF];
#pragma unroll
    for (int k = 0; k < NDOF; k++) a0[k] = fs[k];
    ldl9_solve(L, id, a0);
    double s0 = fsign[0], s1 = fsign[1];
    double r0 = s0 * a0[7] - faref[0], r1 = s1 * a0[8] - faref[1];   // row residuals at zero constraint force
    // with forces f = (f0, f1) >= 0 on the rows: jar_i = r_i + sum_j (s_i W_ij s_j) f_j,  f_i = D_i max(0, -jar_i)
    double G00 = W00 * s0 * s0, G01 = W01 * s0 * s1, G11 = W11 * s1 * s1;
    double f0 = 0, f1 = 0;
    bool have0 = s0 != 0, have1 = s1 != 0, done = false;
    if (have0 && have1) {   // both active: (I + D G) f = -D r
      double a = 1 + fD[0] * G00, b = fD[0] * G01, cc = fD[1] * G01, dd = 1 + fD[1] * G11;
      double det = a * dd - b * cc, y0 = -fD[0] * r0, y1 = -fD[1] * r1;
      double g0 = (dd * y0 - b * y1) / det, g1 = (a * y1 - cc * y0) / det;
      if (g0 > 0 && g1 > 0) { f0 = g0; f1 = g1; done = true; }
    }
    if (!done && have0) {   // only row 0 active
      double g0 = -fD[0] * r0 / (1 + fD[0] * G00);
      if (g0 > 0 && (!have1 || r1 + G01 * g0 >= 0)) { f0 = g0; f1 = 0; done = true; }
    }
    if (!done && have1) {   // only row 1 active
      double g1 = -fD[1] * r1 / (1 + fD[1] * G11);
      if (g1 > 0 && (!have0 || r0 + G01 * g1 >= 0)) { f1 = g1; f0 = 0; done = true; }
    }
    fc[7] = s0 * f0; fc[8] = s1 * f1;
  }
  // Euler with implicit joint damping: (M + h B) qacc = qfrc_smooth + qfrc_constraint  (mj_EulerSkip [ext]).  B acts on
  // the fingers only = the last two dofs, so only the last two pivots and l87 of the LDL^T factors change.
  D3IL_REFRESH(c0, ce);
  {
    double hb0 = ce.timestep * ce.f_damping[0], hb1 = ce.timestep * ce.f_damping[1];
    double l87 = L[tri(8, 7)];
    double S11 = d[8] + l87 * l87 * d[7];      // Schur complement entry (8,8) of M
    double d7n = d[7] + hb0, i7 = rcpd(d7n);
    double l87n = l87 * d[7] * i7;
    double d8n = S11 + hb1 - l87n * l87n * d7n;
    d[7] = d7n; d[8] = d8n; id[7] = i7; id[8] = rcpd(d8n); L[tri(8, 7)] = l87n;
  }
  double qacc[NDOF];
#pragma unroll
  for (int k = 0; k < NDOF; k++) qacc[k] = fs[k] + fc[k];
  ldl9_solve(L, id, qacc);
#pragma unroll
  for (int k = 0; k < NDOF; k++) { st.v[k] += ce.timestep * qacc[k]; st.q[k] += ce.timestep * st.v[k]; }
  if (trig)      // |dq| = h |v| <= 1e-3 x a few rad/s
#pragma unroll
    for (int k = 0; k < NARM; k++) trig_advance(ce.timestep * st.v[k], trig[k], trig[NARM + k]);
}

// joint PD on the IK set-point + finger PD + one physics sub-step (the part of Scene.next_step after the IK update)
template <class C, class R = RareInline>
D3IL_HD void control_and_physics(const C& c, EnvState& st, const double* q_des, const double* qd_des, double set_width, bool grasp, double* warm, double* trig = nullptr, R* rare = nullptr) {
  double tau[NARM], ff[NFING];
#pragma unroll
  for (int k = 0; k < NARM; k++) tau[k] = c.pd_p[k] * (q_des[k] - st.q[k]) + c.pd_d[k] * (qd_des[k] - st.v[k]);
  // RobotBase.fing_ctrl_step (Robots.py:441-476)
  double mean = 0.5 * (st.q[NARM] + st.q[NARM + 1]);
#pragma unroll
  for (int k = 0; k < NFING; k++) {
    double w = st.q[NARM + k], wv = st.v[NARM + k];
    double f1 = 500 * (mean - w), f2;
    if (mean - set_width > 0.005) f2 = grasp ? -20.0 : 10 * (-0.2 - wv);
    else f2 = clampd(500 * (set_width - w) - 10 * wv, -5, 5);
    ff[k] = f1 + f2;
  }
  physics_substep(c, st, tau, ff, warm, trig, rare);
}

// controllers feeding one physics sub-step (Scene.next_step, core/Scene.py:121-138)
template <bool IK, bool FAST, class C>
D3IL_HD void substep(const C& c, EnvState& st, const double* des_pos, const double* des_quat, const double* pd_q, double set_width, bool grasp, double* warm, double* vwarm = nullptr) {
  if (IK) {
    ik_update<FAST>(c, des_pos, des_quat, st.q, st.flags, st.ikq, st.ikqd, vwarm);
    control_and_physics(c, st, st.ikq, st.ikqd, set_width, grasp, warm);
  } else {
    double zero[NARM] = {0, 0, 0, 0, 0, 0, 0};
    control_and_physics(c, st, pd_q, zero, set_width, grasp, warm);
  }
}

// ObstacleAvoidanceEnv.check_mode (avoiding.py:173-202), literal comparisons
template <class C> D3IL_HD void check_mode(const C& c, EnvState& st) {
  auto f = c.task_f;
  double x = st.tcp[0], y = st.tcp[1];
  if (y - 0.03 <= f[0] && f[0] <= y + 0.03 && !(st.flags & F_L1)) {
    if (x < f[4]) st.flags |= 1u << 0; else if (x > f[4]) st.flags |= 1u << 1;
    st.flags |= F_L1;
  }
  if (y - 0.03 <= f[1] && f[1] <= y + 0.03 && !(st.flags & F_L2)) {
    if (x < f[5]) st.flags |= 1u << 2; else if (f[5] < x && x < f[6]) st.flags |= 1u << 3; else if (x > f[6]) st.flags |= 1u << 4;
    st.flags |= F_L2;
  }
  if (y >= f[2] && !(st.flags & F_L3)) {
    if (x < f[7]) st.flags |= 1u << 5;
    if (f[7] < x && x < f[8]) st.flags |= 1u << 6; else if (f[8] < x && x < f[9]) st.flags |= 1u << 7; else if (x > f[7]) st.flags |= 1u << 8;
    st.flags |= F_L3;
  }
}

// GymEnvWrapper.step before the physics (gyms/gym_env_wrapper.py:88-90): observation and is_finished() of the state
// produced by the PREVIOUS call; ObstacleAvoidanceEnv._check_early_termination (avoiding.py:236-246)
template <class C> D3IL_HD void step_begin(const C& c, EnvState& st, float* obs, unsigned char* done, int max_steps) {
  obs[0] = (float)st.tcp[0]; obs[1] = (float)st.tcp[1];
  bool fin = (st.flags & F_TERMINATED) != 0;
  if (!fin) {
    bool succ = st.tcp[1] > c.task_f[3];
    if (succ || (st.flags & F_ROD_CONTACT)) { if (succ) st.flags |= F_SUCCESS; st.flags |= F_TERMINATED; fin = true; }
  }
  if (!fin && st.step >= max_steps - 1) fin = true;
  *done = fin ? 1 : 0;
}
// controller.setSetPoint(action) (IKControllers.py:346-362): position + normalised quaternion
// A NaN / Inf action (a diverged policy) must not reach the dynamics: integer test of the exponent field, which
// -ffinite-math-only cannot fold away.  A bad action is replaced by a fixed reachable set-point; the caller flags the lane
// (F_SOLVER_FAIL | F_TERMINATED).  The reference has no such check (MuJoCo would warn and reset its data on the NaN qacc).
// The test reads the action words from MEMORY as integers (memcpy from the source pointer): tested on an already loaded double, the
// exponent pattern is recognised as an fp-class test and -ffinite-math-only folds its NaN half away (seen in the Stacking kernel:
// Inf was caught, NaN was not).
D3IL_HD bool action_is_bad(const double* __restrict__ a, int count = 7) {
  bool bad = false;
  for (int k = 0; k < count; k++) { unsigned long long b; __builtin_memcpy(&b, a + k, 8); bad = bad || ((b >> 52) & 0x7ffull) == 0x7ffull; }
  return bad;
}
// act: the action as loaded, mem: where it was loaded from
D3IL_HD bool sanitize_action(double* act, const double* __restrict__ mem) {
  const bool bad = action_is_bad(mem);
  if (bad) { act[0] = 0.5; act[1] = 0.0; act[2] = 0.3; act[3] = 0.0; act[4] = 1.0; act[5] = 0.0; act[6] = 0.0; }
  return bad;
}
D3IL_HD void make_setpoint(const double* action, double* des) {
  double n = sqrt(action[3] * action[3] + action[4] * action[4] + action[5] * action[5] + action[6] * action[6]);
  des[0] = action[0]; des[1] = action[1]; des[2] = action[2];
  des[3] = action[3] / n; des[4] = action[4] / n; des[5] = action[5] / n; des[6] = action[6] / n;
}
template <class C> D3IL_HD void step_end(const C& c, EnvState& st) {
  st.step += 1;
  check_mode(c, st);
}

// ObstacleAvoidanceEnv.step (avoiding.py:168-171) over GymEnvWrapper.step (gyms/gym_env_wrapper.py:45-100)
template <bool FAST, class C>
D3IL_HD void env_step(const C& c, EnvState& st, const double* action, float* obs, unsigned char* done, int n_substeps, int max_steps) {
  step_begin(c, st, obs, done, max_steps);
  double des[7];
  make_setpoint(action, des);
  double warm[6], vwarm[7];
  warm[5] = 0.0; vwarm[6] = 0.0;
#pragma clang loop unroll(disable)
  for (int s = 0; s < n_substeps; s++) {
    D3IL_REFRESH(c, cs);
    substep<true, FAST>(cs, st, des, des + 3, nullptr, 0.04, false, warm, vwarm);
  }
  step_end(c, st);
}

// ObstacleAvoidanceEnv.reset (avoiding.py:248-262): scene.reset, beam to init_qpos, one PD-hold sub-step
template <class C> D3IL_HD void env_reset(const C& c, EnvState& st, const double* init_qpos, float* obs) {
#pragma unroll
  for (int k = 0; k < NARM; k++) { st.q[k] = init_qpos[k]; st.ikq[k] = 0; st.ikqd[k] = 0; }
  st.q[NARM] = 0; st.q[NARM + 1] = 0;
#pragma unroll
  for (int k = 0; k < NDOF; k++) st.v[k] = 0;
  st.flags = 0; st.step = 0;
  // mj_forward at the beamed state (MjScene.set_state, MjScene.py:294-299): qfrc_bias used by the first command
  DynOut dyn;
  dynamics(c, st.q, st.v, dyn);
#pragma unroll
  for (int k = 0; k < NARM; k++) st.bias[k] = dyn.bias[k];
  double warm[6];
  warm[5] = 0.0;
  substep<false, true>(c, st, nullptr, nullptr, init_qpos, 0.001, false, warm);
  obs[0] = (float)st.tcp[0]; obs[1] = (float)st.tcp[1];
}

// ------------------------------------------------------------------ host: finish the constant block
#if defined(__HIPCC__)
#define D3IL_HOST __host__
#else
#define D3IL_HOST
#endif
D3IL_HOST inline void finish_invweights(PandaConsts& c) {
  double q[NDOF] = {0}, v[NDOF] = {0};
  DynOut dyn;
  dynamics(c, q, v, dyn);
  double L[45], d[NDOF], id[NDOF], Minv[NDOF][NDOF];
  ldl9(dyn.M, L, d, id);
  for (int col = 0; col < NDOF; col++) {
    double e[NDOF] = {0}; e[col] = 1; ldl9_solve(L, id, e);
    for (int r = 0; r < NDOF; r++) Minv[r][col] = e[r];
  }
  for (int k = 0; k < NDOF; k++) c.dof_invweight0[k] = Minv[k][k];
  // translational inverse weight of the rod body at its COM (= rod geom centre): mean diag of J Minv J^T
  double sn[NARM], cs[NARM], R7[9], p7[3], ax[NARM][3], og[NARM][3];
  for (int i = 0; i < NARM; i++) { sn[i] = 0; cs[i] = 1; }
  world_chain(c, sn, cs, R7, p7, ax, og);
  double pc[3]; mulE(R7, c.rod_c7, pc); pc[0] += p7[0]; pc[1] += p7[1]; pc[2] += p7[2];
  double J[3][NARM];
  for (int k = 0; k < NARM; k++) { double dd[3] = {pc[0] - og[k][0], pc[1] - og[k][1], pc[2] - og[k][2]}, col[3]; cross3(ax[k], dd, col); J[0][k] = col[0]; J[1][k] = col[1]; J[2][k] = col[2]; }
  double tr = 0;
  for (int r = 0; r < 3; r++) for (int a = 0; a < NARM; a++) for (int b = 0; b < NARM; b++) tr += J[r][a] * Minv[a][b] * J[r][b];
  c.rod_invweight0 = fmax(1e-15, tr / 3);
}

}  // namespace d3il
