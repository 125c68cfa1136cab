// push_step.h - per-environment math of the Pushing task (Block_Push_Env, envs/gym_pushing_env/gym_pushing/envs/pushing.py):
// the Avoiding arm (panda_step.h: IK controller, joint PD, arm dynamics) plus two free 6 cm cubes on the table, pushed by
// the rod.  One lane owns one environment.  Everything here compiles for the host as well (tests/hostcheck).
//
// What one physics sub-step adds on top of panda_step.h (mj_step of the 21-dof model, MjScene.py:110-111):
//   * free-body kinematics of the cubes (isotropic inertia => no gyroscopic bias), gravity only;
//   * collision: cube <-> table_plane slab, cube <-> support slab, cube <-> cube (SAT + face clipping), rod <-> cube
//     (axis segment - box distance minus the radius);
//   * soft-constraint solve over all 21 dofs: 9 joint-limit rows + elliptic friction cones (condim 3) of every
//     contact.  Primal Newton with exact line search on  1/2 (a-a0)'M(a-a0) + sum_i s_i(J_i a - aref_i)  (strictly
//     convex, so the optimum is the one MuJoCo's Newton solver converges to).  The Hessian lives in LDS
//     (lower-triangular, 231 entries per lane, lane-strided so accesses are bank-conflict free); its Cholesky factor is
//     computed inside the skyline of the block structure [cube1 | cube2 | arm] (cube-cube coupling exists only with a
//     cube-cube contact, arm-cube coupling only with a rod contact), decided per wave;
//   * semi-implicit Euler: cubes integrate position and quaternion (angular velocity in body axes), the arm keeps the
//     implicit finger damping of panda_step.h.
//
// Contact records, the arm mass matrix and the solver vectors live in a per-lane scratch area in HBM (structure of
// arrays, lane-strided => coalesced), which the 4 MB L2 of each XCD holds entirely at the benchmark sizes.
#pragma once
#include "panda_step.h"

#if defined(D3IL_DEVICE_STATS) && defined(__HIP_DEVICE_COMPILE__)
#define PUSH_TIC unsigned long long t_tic_ = wall_clock64()
#define PUSH_TOC(slot) do { unsigned long long t_now_ = wall_clock64(); \
    if (__builtin_amdgcn_mbcnt_hi(__builtin_amdgcn_read_exec_hi(), __builtin_amdgcn_mbcnt_lo(__builtin_amdgcn_read_exec_lo(), 0u)) == 0 && blockIdx.x < 4096) \
      atomicAdd(&d3il::g_dev_wave[blockIdx.x][slot], t_now_ - t_tic_); t_tic_ = t_now_; } while (0)
// wave-level event counter of the counting build (-DD3IL_DEVICE_STATS -DD3IL_DEVICE_COUNTS; the atomics inside the contact loops distort the timers,
// so the plain stats build leaves them out): +1 per wave (its first active lane) each time the statement is reached
#if defined(D3IL_DEVICE_COUNTS)
#define PUSH_CNT(slot) do { if (__builtin_amdgcn_mbcnt_hi(__builtin_amdgcn_read_exec_hi(), __builtin_amdgcn_mbcnt_lo(__builtin_amdgcn_read_exec_lo(), 0u)) == 0 && blockIdx.x < 4096) \
      atomicAdd(&d3il::g_dev_cnt[blockIdx.x][slot], 1ull); } while (0)
#else
#define PUSH_CNT(slot) ((void)0)
#endif
#else
#define PUSH_TIC ((void)0)
#define PUSH_TOC(slot) ((void)0)
#define PUSH_CNT(slot) ((void)0)
#endif

namespace d3il {

constexpr int PUSH_NB = 2;              // free cubes
constexpr int PUSH_NV = 21;             // solver dof order: cube1[6] cube2[6] arm[9]
constexpr int PUSH_ARM0 = 12;
constexpr int PUSH_MAXCON = 24;
constexpr int PUSH_NH = PUSH_NV * (PUSH_NV + 1) / 2;   // 231
constexpr int PUSH_MAXIT = 40;
// Stopping rule of the contact Newton solvers (Pushing and the generic engine), a run-time setting so that the device path can be
// run with the oracle's rule for parity A/B tests (d3il_set_option "solver_strict"; tests/test_gpu_parity_*).
//   grad_tol : gradient (generalised force, N / N m) below which an iterate is accepted without a further Newton step
//              (MuJoCo: scaled gradient below `tolerance` = 1e-10)
//   step_rel : an accepted full Newton step below step_rel (relative) ends the iteration (quadratic convergence leaves its square)
//   ls_c2    : curvature condition of the line search |phi'(alpha)| <= ls_c2 |phi'(0)|
//   ls_full  : the full step is taken when phi'(1) <= ls_full |phi'(0)|
//   ls_rel   : or when the minimiser of phi is within ls_rel (relative) of alpha
struct SolverTol { double grad_tol, step_rel, ls_c2, ls_full, ls_rel; };
constexpr SolverTol SOLVER_TOL_PRODUCTION = {1e-10, 1e-6, D3IL_LS_C2, 0.1, 1e-3};
constexpr SolverTol SOLVER_TOL_STRICT = {1e-13, 1e-10, 1e-6, 1e-6, 1e-9};     // the oracle iterates to round-off (scaled gradient 1e-15, exact line search)
#if defined(__HIPCC__)
__constant__ SolverTol g_solver_tol = {1e-10, 1e-6, D3IL_LS_C2, 0.1, 1e-3};
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define D3IL_TOL g_solver_tol
#else
inline SolverTol& host_solver_tol() { static SolverTol t = SOLVER_TOL_PRODUCTION; return t; }
#define D3IL_TOL host_solver_tol()
#endif
#define PUSH_GRAD_TOL (D3IL_TOL.grad_tol)

// f64 state fields per environment in the SoA state buffer: the 42 arm fields of Avoiding (D3IL_STATE_*), then per cube
// pos[3] quat[4] vel[6] (linear world, angular body axes = MuJoCo free-joint qvel), then the solver warm start qacc[21]
constexpr int PUSH_STATE_BOX = 42;
constexpr int PUSH_STATE_WARM = PUSH_STATE_BOX + 13 * PUSH_NB;
constexpr int PUSH_STATE_F64 = PUSH_STATE_WARM + PUSH_NV;   // 89

// flag bits (EnvState::flags).  F_TERMINATED / F_SUCCESS / F_IK_VALID / F_SOLVER_FAIL keep their Avoiding positions.
enum : unsigned {
  PF_FIRST_MASK = 0x7u,          // first_visit + 1   (pushing.py:341-377)
  PF_MODE_SHIFT = 3, PF_MODE_MASK = 0x7u << 3,   // mode + 1
  PF_WARM_VALID = 1u << 6,
  PF_CON_OVERFLOW = 1u << 18,    // more than PUSH_MAXCON contacts in one sub-step (extra contacts dropped)
  PF_OFF_TABLE = 1u << 19,       // a cube left the modelled part of the table top
};

struct PushConsts {
  double box_half[3];
  double box_mass, box_inertia;          // isotropic (cube)
  double box_invw_t, box_invw_r;         // body_invweight0 of a cube
  double slab_c[2][3], slab_h[2][3];     // table_plane, support_body (axis aligned, static)
  // contact parameter sets after mj_contactParam mixing: 0 = cube-slab, 1 = cube-cube and rod-cube
  double ct_K[2], ct_B[2], ct_solimp[2][5], ct_fric[2];
  double target[2][3], min_dist;         // pushing_objects.py:11-15, pushing.py:251
  double impratio;
};

// On the device the constants sit in constant memory and every function re-binds its `pc` to that object, so that no generic
// pointer to them survives a (non-inlined) call boundary.  One Pushing model per process.
#if defined(__HIPCC__)
__constant__ PushConsts g_push_consts;
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define D3IL_PUSH_CONSTS(in, name) const PushConsts& name = g_push_consts; (void)in
#else
#define D3IL_PUSH_CONSTS(in, name) const PushConsts& name = in
#endif

struct BoxState { double pos[3], quat[4], vel[6]; };
struct PushState {
  EnvState arm;
  BoxState box[PUSH_NB];
};

// per-lane views of the scratch areas: h = coupled-solver table (LDS on the device), g = memory-resident solver (HBM), w =
// warm start.  On the device the pointers carry their address space so that accesses compile to ds_* / global_*
// instructions instead of flat ones, and the LDS lane stride is a compile-time constant.
#ifndef D3IL_PUSH_LANES
#define D3IL_PUSH_LANES 24
#endif
constexpr int PUSH_LANES = D3IL_PUSH_LANES;   // environments per workgroup (push_kernels.h)
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(3))) double push_lds_double;
typedef __attribute__((address_space(1))) double push_glb_double;
#define PUSH_HS PUSH_LANES
#else
typedef double push_lds_double;
typedef double push_glb_double;
#define PUSH_HS 1
#endif
struct PushScratch {
  push_lds_double* h;
  push_glb_double* g; int gs;
  push_glb_double* w; int ws;     // warm start qacc[21] of the constraint solver: rows PUSH_STATE_WARM.. of the state buffer
};
// layout of the g area (doubles per lane)
constexpr int PG_M = 0;                                 // arm mass matrix, packed lower 9x9
constexpr int PG_CON = PG_M + 45;                       // contact records
constexpr int PREC = 26;
//   record: 0 pos[3] | 3 frame[9] (normal, t1, t2) | 12 dist | 13 kind | 14 cube | 15 aref[3] | 18 Dn | 19 mu | 20 jar[3] | 23 Jp[3]
constexpr int PG_JA = PG_CON + PUSH_MAXCON * PREC;      // arm Jacobian rows of the (up to two) rod contacts: 2 x 3 x 7
constexpr int PG_A0 = PG_JA + 42;                       // qacc_smooth[21]
constexpr int PG_X = PG_A0 + PUSH_NV;                   // iterate
constexpr int PG_GRAD = PG_X + PUSH_NV;                 // gradient, then the Newton direction
constexpr int PG_H = PG_GRAD + PUSH_NV + 72;            // after the PG_AUX block: Hessian of the slow general path (231)
constexpr int PG_SIZE = PG_H + PUSH_NH;

#define PGS(i) sc.g[(long)(i) * sc.gs]
#define PHS(i) PGS(PG_H + (i))
#define PTS(i) sc.h[(i) * PUSH_HS]
#define PWS(i) sc.w[(long)(i) * sc.ws]

enum { CK_SLAB = 0, CK_BOXBOX = 1, CK_ROD = 2 };

D3IL_HD void quat2mat(const double* q, double* R) {   // mju_quat2Mat [ext]
  double q00 = q[0] * q[0], q11 = q[1] * q[1], q22 = q[2] * q[2], q33 = q[3] * q[3];
  double q01 = q[0] * q[1], q02 = q[0] * q[2], q03 = q[0] * q[3], q12 = q[1] * q[2], q13 = q[1] * q[3], q23 = q[2] * q[3];
  R[0] = q00 + q11 - q22 - q33; R[4] = q00 - q11 + q22 - q33; R[8] = q00 - q11 - q22 + q33;
  R[1] = 2 * (q12 - q03); R[2] = 2 * (q13 + q02); R[3] = 2 * (q12 + q03);
  R[5] = 2 * (q23 - q01); R[6] = 2 * (q13 - q02); R[7] = 2 * (q23 + q01);
}

// ------------------------------------------------------------------------------------------------ collision
// box-box: separating-axis test over 15 axes, then a face contact (incident face clipped against the reference face;
// every clipped vertex inside the margin is a contact, positioned midway between the surfaces) or one edge-edge contact.
// out[k] = {dist, pos[3], normal[3]}, normal from box 1 to box 2.  p: centres, R: row-major rotation (columns = axes).
// Register-only formulation: every array index is a compile-time constant after unrolling (dynamic axis choices are resolved with
// selects, the Sutherland-Hodgman clip grows its polygon by select-chain inserts), so the routine needs no private (scratch) memory
// on the device.  Contacts are handed to `emit(dist, pos[3], normal[3])` in polygon order; at most `cap` (<= 8) are emitted.
// Device: the three candidates pass through an opaque move before the select.  Without it the compiler turns "select of loaded values"
// back into "load from a selected address", which keeps the source arrays in private memory (dynamic scratch indexing).
#if defined(__HIP_DEVICE_COMPILE__)
#define D3IL_OPAQUE3(a, b, c) asm volatile("" : "+v"(a), "+v"(b), "+v"(c))
#else
#define D3IL_OPAQUE3(a, b, c) ((void)0)
#endif
D3IL_HD double bb_sel3(const double* v, int i) {
  double a = v[0], b = v[1], c = v[2];
  D3IL_OPAQUE3(a, b, c);
  return i == 0 ? a : (i == 1 ? b : c);
}
D3IL_HD void bb_row(const double (*M)[3], int i, double* o) {
#pragma unroll
  for (int k = 0; k < 3; k++) { double a = M[0][k], b = M[1][k], c = M[2][k]; D3IL_OPAQUE3(a, b, c); o[k] = i == 0 ? a : (i == 1 ? b : c); }
}
template <int N> D3IL_HD void bb_put(double (*P)[3], int at, double x, double y, double z) {      // P[at] = (x, y, z), at < N
#pragma unroll
  for (int j = 0; j < N; j++) if (at == j) { P[j][0] = x; P[j][1] = y; P[j][2] = z; }
}
// one side of the clip: IN vertices in, at most IN + 1 out (a convex polygon gains at most one vertex per half plane)
template <int IN> D3IL_HD int bb_clip_side(const double (*poly)[3], int np, int cdim, double sg, double lim, double etol, double (*out)[3]) {
  int nn = 0;
#pragma unroll
  for (int v = 0; v < IN; v++) {
    if (v >= np) continue;
    double q[3];      // successor vertex: v + 1, or vertex 0 after the last one
#pragma unroll
    for (int k = 0; k < 3; k++) q[k] = (v + 1 == np || v + 1 >= IN) ? poly[0][k] : poly[v + 1 < IN ? v + 1 : 0][k];
    const double pc = cdim ? poly[v][1] : poly[v][0], qc = cdim ? q[1] : q[0];
    const double fp = sg * pc - lim, fq = sg * qc - lim;
    if (fp <= etol) { bb_put<IN + 1>(out, nn, poly[v][0], poly[v][1], poly[v][2]); nn++; }
    if ((fp <= etol) != (fq <= etol)) {
      const double t = fp / (fp - fq);
      bb_put<IN + 1>(out, nn, poly[v][0] + t * (q[0] - poly[v][0]), poly[v][1] + t * (q[1] - poly[v][1]), poly[v][2] + t * (q[2] - poly[v][2]));
      nn++;
    }
    if (nn > IN) nn = IN + 1;     // cannot happen for a convex polygon; keeps the inserts inside the array under round-off
  }
  return nn > IN + 1 ? IN + 1 : nn;
}
template <class EMIT>
D3IL_HD int box_box_emit(const double* p1, const double* R1, const double* s1, const double* p2, const double* R2, const double* s2,
                         double margin, int cap, EMIT emit) {
  const double FUDGE = 1.05;
  const double ETOL = 1e-12;   // a vertex this close to a side plane of the reference face counts as inside (faces of equal extent lying on each other)
  double A[3][3], B[3][3], d[3], Cm[3][3], Q[3][3], dA[3], dB[3];
#pragma unroll
  for (int i = 0; i < 3; i++) {
#pragma unroll
    for (int k = 0; k < 3; k++) { A[i][k] = R1[3 * k + i]; B[i][k] = R2[3 * k + i]; }
  }
#pragma unroll
  for (int k = 0; k < 3; k++) d[k] = p2[k] - p1[k];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    dA[i] = dot3(d, A[i]); dB[i] = dot3(d, B[i]);
#pragma unroll
    for (int j = 0; j < 3; j++) { Cm[i][j] = dot3(A[i], B[j]); Q[i][j] = fabs(Cm[i][j]); }
  }
  double best = -1e300; int code = -1; double nsign = 1;
  bool apart = false;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    double sep = fabs(dA[i]) - (s1[i] + s2[0] * Q[i][0] + s2[1] * Q[i][1] + s2[2] * Q[i][2]);
    if (sep > margin) apart = true;
    if (!apart && sep > best + 1e-10) { best = sep; code = i; nsign = dA[i] < 0 ? -1 : 1; }
  }
  if (apart) return 0;
#pragma unroll
  for (int j = 0; j < 3; j++) {
    double sep = fabs(dB[j]) - (s2[j] + s1[0] * Q[0][j] + s1[1] * Q[1][j] + s1[2] * Q[2][j]);
    if (sep > margin) apart = true;
    // a face axis of box 2 must beat box 1's by more than 1e-10 - unless the two tie within that band and box 2 offers the larger
    // face: a small box lying flat on a big one is then clipped against the big face (reference) whichever geom comes first
    bool wins = sep > best + 1e-10;
    const int c1 = code < 3 ? (code + 1) % 3 : 0, c2 = code < 3 ? (code + 2) % 3 : 0;
    if (!wins && code >= 0 && code < 3 && sep >= best - 1e-10 && s2[(j + 1) % 3] * s2[(j + 2) % 3] > bb_sel3(s1, c1) * bb_sel3(s1, c2)) wins = true;
    if (!apart && wins) { best = sep; code = 3 + j; nsign = dB[j] < 0 ? -1 : 1; }
  }
  if (apart) return 0;
  double en[3] = {0, 0, 0};
#pragma unroll
  for (int i = 0; i < 3; i++) {
#pragma unroll
    for (int j = 0; j < 3; j++) {
      constexpr int M3[5] = {0, 1, 2, 0, 1};
      const int i1 = M3[i + 1], i2 = M3[i + 2], j1 = M3[j + 1], j2 = M3[j + 2];
      double l2 = 1 - Cm[i][j] * Cm[i][j];
      if (l2 < 1e-10 || apart) continue;
      double l = sqrt(l2);
      double proj = dA[i2] * Cm[i1][j] - dA[i1] * Cm[i2][j];
      double ra = s1[i1] * Q[i2][j] + s1[i2] * Q[i1][j], rb = s2[j1] * Q[i][j2] + s2[j2] * Q[i][j1];
      double sep = (fabs(proj) - (ra + rb)) / l;
      if (sep > margin) { apart = true; continue; }
      // 5 % better: a shallower penetration (sep < 0) or, for boxes apart but inside the margin (sep > 0), a larger gap
      if (sep > 0 ? sep > best * FUDGE + 1e-10 : (sep * FUDGE > best + 1e-10 && sep > best)) {
        best = sep; code = 6 + 3 * i + j;
        double Lx[3]; cross3(A[i], B[j], Lx);
        double sg = proj < 0 ? -1 : 1;
#pragma unroll
        for (int k = 0; k < 3; k++) en[k] = sg * Lx[k] / l;
      }
    }
  }
  if (apart) return 0;
  if (code >= 6) {
    const int i = (code - 6) / 3, j = (code - 6) % 3;
    double pa[3], pb[3], Ai_[3], Bj_[3];
    bb_row(A, i, Ai_); bb_row(B, j, Bj_);
#pragma unroll
    for (int k = 0; k < 3; k++) { pa[k] = p1[k]; pb[k] = p2[k]; }
#pragma unroll
    for (int a = 0; a < 3; a++) if (a != i) { double sg = dot3(en, A[a]) > 0 ? 1 : -1;
#pragma unroll
      for (int k = 0; k < 3; k++) pa[k] += sg * s1[a] * A[a][k]; }
#pragma unroll
    for (int b = 0; b < 3; b++) if (b != j) { double sg = dot3(en, B[b]) > 0 ? -1 : 1;
#pragma unroll
      for (int k = 0; k < 3; k++) pb[k] += sg * s2[b] * B[b][k]; }
    double w[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
    double crow[3] = {bb_sel3(Cm[0], j), bb_sel3(Cm[1], j), bb_sel3(Cm[2], j)};
    double cc = bb_sel3(crow, i), wa = dot3(w, Ai_), wb = dot3(w, Bj_), den = 1 - cc * cc;
    double al = (wa - cc * wb) / den, be = (cc * wa - wb) / den;
    if (cap < 1) return 0;
    double pos[3];
#pragma unroll
    for (int k = 0; k < 3; k++) pos[k] = 0.5 * (pa[k] + al * Ai_[k] + pb[k] + be * Bj_[k]);
    emit(best, pos, en);
    return 1;
  }
  const bool ref2 = code >= 3; const int ax = ref2 ? code - 3 : code;
  double pr[3], pi[3], sr[3], si[3], Ar[3][3], Ai[3][3];
#pragma unroll
  for (int k = 0; k < 3; k++) { pr[k] = ref2 ? p2[k] : p1[k]; pi[k] = ref2 ? p1[k] : p2[k]; sr[k] = ref2 ? s2[k] : s1[k]; si[k] = ref2 ? s1[k] : s2[k]; }
#pragma unroll
  for (int a = 0; a < 3; a++) {
#pragma unroll
    for (int k = 0; k < 3; k++) { Ar[a][k] = ref2 ? B[a][k] : A[a][k]; Ai[a][k] = ref2 ? A[a][k] : B[a][k]; }
  }
  const int a1 = ax == 2 ? 0 : ax + 1, a2 = ax == 0 ? 2 : ax - 1;      // (ax + 1) % 3, (ax + 2) % 3
  double Arx[3], Ar1[3], Ar2[3];
  bb_row(Ar, ax, Arx); bb_row(Ar, a1, Ar1); bb_row(Ar, a2, Ar2);
  const double srx = bb_sel3(sr, ax), sr1 = bb_sel3(sr, a1), sr2 = bb_sel3(sr, a2);
  double n[3]; const double sgn = ref2 ? -nsign : nsign;
#pragma unroll
  for (int k = 0; k < 3; k++) n[k] = sgn * Arx[k];
  int kin = 0; double bestdot = -1;
#pragma unroll
  for (int k = 0; k < 3; k++) { double t = fabs(dot3(n, Ai[k])); if (t > bestdot) { bestdot = t; kin = k; } }
  const int k1 = kin == 2 ? 0 : kin + 1, k2 = kin == 0 ? 2 : kin - 1;
  double Aik[3], Ai1[3], Ai2[3];
  bb_row(Ai, kin, Aik); bb_row(Ai, k1, Ai1); bb_row(Ai, k2, Ai2);
  const double sik = bb_sel3(si, kin), si1 = bb_sel3(si, k1), si2 = bb_sel3(si, k2);
  const double sgi = dot3(n, Aik) > 0 ? -1 : 1;
  double p4[4][3];
  bool inside = true;      // the incident face lies within the side planes of the reference face: nothing to clip
#pragma unroll
  for (int v = 0; v < 4; v++) {
    double c0 = (v == 0 || v == 3) ? 1.0 : -1.0, c1 = v < 2 ? 1.0 : -1.0, x[3];
#pragma unroll
    for (int k = 0; k < 3; k++) x[k] = pi[k] + sgi * sik * Aik[k] + c0 * si1 * Ai1[k] + c1 * si2 * Ai2[k] - pr[k];
    p4[v][0] = dot3(x, Ar1); p4[v][1] = dot3(x, Ar2); p4[v][2] = dot3(x, n) - srx;
    inside = inside && p4[v][0] - sr1 <= ETOL && -p4[v][0] - sr1 <= ETOL && p4[v][1] - sr2 <= ETOL && -p4[v][1] - sr2 <= ETOL;
  }
  auto out_vertex = [&](double px, double py, double w) {
    double pos[3], nn[3];
#pragma unroll
    for (int k = 0; k < 3; k++) { pos[k] = pr[k] + px * Ar1[k] + py * Ar2[k] + (srx + 0.5 * w) * n[k]; nn[k] = ref2 ? -n[k] : n[k]; }
    emit(w, pos, nn);
  };
  int cnt = 0;
  if (inside) {
    // the clipping below would return the four vertices unchanged and in order; they are distinct, so no duplicate test either
#pragma unroll
    for (int v = 0; v < 4; v++) {
      const double w = p4[v][2];
      if (w >= margin || cnt >= cap) continue;
      out_vertex(p4[v][0], p4[v][1], w);
      cnt++;
    }
    return cnt;
  }
  // Sutherland-Hodgman against the four side planes of the reference face (side: +a1, -a1, +a2, -a2)
  double q5[5][3], q6[6][3], q7[7][3], poly[8][3];
  int np = bb_clip_side<4>(p4, 4, 0, 1.0, sr1, ETOL, q5);
  if (np == 0) return 0;
  np = bb_clip_side<5>(q5, np, 0, -1.0, sr1, ETOL, q6);
  if (np == 0) return 0;
  np = bb_clip_side<6>(q6, np, 1, 1.0, sr2, ETOL, q7);
  if (np == 0) return 0;
  np = bb_clip_side<7>(q7, np, 1, -1.0, sr2, ETOL, poly);
  if (np == 0) return 0;
#pragma unroll
  for (int v = 0; v < 8; v++) {
    if (v >= np || cnt >= cap) continue;
    const double w = poly[v][2];
    if (w >= margin) continue;
    bool dup = false;
#pragma unroll
    for (int q = 0; q < v; q++) if (fabs(poly[q][0] - poly[v][0]) + fabs(poly[q][1] - poly[v][1]) < 1e-12 && poly[q][2] < margin) dup = true;
    if (dup) continue;
    out_vertex(poly[v][0], poly[v][1], w);
    cnt++;
  }
  return cnt;
}
// array interface (host build, reset kernels, the Pushing / Sorting engines): out[k] = {dist, pos[3], normal[3]}
D3IL_HD int box_box(const double* p1, const double* R1, const double* s1, const double* p2, const double* R2, const double* s2,
                    double margin, double (*out)[7], int cap) {
  int n = 0;
  return box_box_emit(p1, R1, s1, p2, R2, s2, margin, cap < 8 ? cap : 8, [&](double dist, const double* pos, const double* nrm) {
    out[n][0] = dist;
    for (int k = 0; k < 3; k++) { out[n][1 + k] = pos[k]; out[n][4 + k] = nrm[k]; }
    n++;
  });
}

// rod (cylinder, axis u through pc, radius rad, half length half) against a box: closest points of the axis segment
// and the box, minus the radius (side contacts; the flat end caps are not modelled).  Normal from the box to the rod.
D3IL_HD bool cyl_box(const double* pc, const double* axis, double rad, double half, const double* pb, const double* Rb, const double* sb,
                     double margin, double* out) {
  double c[3], u[3], rel[3] = {pc[0] - pb[0], pc[1] - pb[1], pc[2] - pb[2]};
  for (int i = 0; i < 3; i++) { double col[3] = {Rb[i], Rb[3 + i], Rb[6 + i]}; c[i] = dot3(rel, col); u[i] = dot3(axis, col); }
  // candidate parameters: the two ends and the (up to six) crossings of the box's face planes; invalid ones collapse onto +half.
  // Sorted with a fixed 19-comparator network and scanned with unrolled loops: everything stays in registers.
  double T[8];
  T[0] = -half; T[7] = half;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    bool okc = fabs(u[i]) > 1e-14;
    double iu = okc ? 1.0 / u[i] : 0.0;
    double ta = (sb[i] - c[i]) * iu, tb = (-sb[i] - c[i]) * iu;
    T[1 + 2 * i] = (okc && ta > -half && ta < half) ? ta : half;
    T[2 + 2 * i] = (okc && tb > -half && tb < half) ? tb : half;
  }
#define PUSH_CE(a, b) { double lo_ = fmin(T[a], T[b]), hi_ = fmax(T[a], T[b]); T[a] = lo_; T[b] = hi_; }
  PUSH_CE(0, 1) PUSH_CE(2, 3) PUSH_CE(4, 5) PUSH_CE(6, 7) PUSH_CE(0, 2) PUSH_CE(1, 3) PUSH_CE(4, 6) PUSH_CE(5, 7) PUSH_CE(1, 2) PUSH_CE(5, 6)
  PUSH_CE(0, 4) PUSH_CE(3, 7) PUSH_CE(1, 5) PUSH_CE(2, 6) PUSH_CE(1, 4) PUSH_CE(3, 6) PUSH_CE(2, 4) PUSH_CE(3, 5) PUSH_CE(3, 4)
#undef PUSH_CE
  double G[8];
#pragma unroll
  for (int k = 0; k < 8; k++) {
    double g = 0;
#pragma unroll
    for (int i = 0; i < 3; i++) { double x = c[i] + T[k] * u[i], cl = x > sb[i] ? sb[i] : (x < -sb[i] ? -sb[i] : x); g += u[i] * (x - cl); }
    G[k] = g;
  }
  const double tol = 1e-13;
  double tm = T[7], tp = T[0];
  {   // smallest t with g(t) >= 0: first candidate with G >= -tol, interpolated from its predecessor when it is beyond the root
    bool found = false;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      bool hit = !found && !(G[k] < -tol);
      if (hit) tm = (k == 0 || !(G[k] > tol)) ? T[k] : T[k > 0 ? k - 1 : 0] + (T[k] - T[k > 0 ? k - 1 : 0]) * (-G[k > 0 ? k - 1 : 0]) / (G[k] - G[k > 0 ? k - 1 : 0]);
      found = found || hit;
    }
  }
  {   // largest t with g(t) <= 0, scanning from the upper end
    bool found = false;
#pragma unroll
    for (int k = 7; k >= 0; k--) {
      bool hit = !found && !(G[k] > tol);
      if (hit) tp = (k == 7 || !(G[k] < -tol)) ? T[k] : T[k] + (T[k < 7 ? k + 1 : 7] - T[k]) * (-G[k]) / (G[k < 7 ? k + 1 : 7] - G[k]);
      found = found || hit;
    }
  }
  double ts = 0.5 * (tm + tp), x[3], q[3], df[3], len = 0;
  for (int i = 0; i < 3; i++) { x[i] = c[i] + ts * u[i]; q[i] = x[i] > sb[i] ? sb[i] : (x[i] < -sb[i] ? -sb[i] : x[i]); df[i] = x[i] - q[i]; len += df[i] * df[i]; }
  len = sqrt(len);
  double nl[3], dist;
  if (len > 1e-9) { for (int i = 0; i < 3; i++) nl[i] = df[i] / len; dist = len - rad; }
  else {
    int bi = 0; double bd = 1e300;
    for (int i = 0; i < 3; i++) { double dd = sb[i] - fabs(x[i]); if (dd < bd) { bd = dd; bi = i; } }
    nl[0] = nl[1] = nl[2] = 0; nl[bi] = x[bi] < 0 ? -1 : 1; dist = -bd - rad;
    q[bi] = nl[bi] * sb[bi];
  }
  if (dist >= margin) return false;
  out[0] = dist;
  for (int k = 0; k < 3; k++) {
    double pw = 0, nw = 0;
    for (int i = 0; i < 3; i++) { pw += Rb[3 * k + i] * (q[i] + 0.5 * dist * nl[i]); nw += Rb[3 * k + i] * nl[i]; }
    out[1 + k] = pb[k] + pw; out[4 + k] = nw;
  }
  return true;
}

// ------------------------------------------------------------------------------------------------ constraint rows
// Jacobian row of a cube (6 entries: linear world, angular body axes) for a world direction f and the arm r = p - centre
D3IL_HD void box_row_r(const double* R, const double* r, const double* f, double* row) {
  double rxf[3];
  cross3(r, f, rxf);
  row[0] = f[0]; row[1] = f[1]; row[2] = f[2];
  row[3] = R[0] * rxf[0] + R[3] * rxf[1] + R[6] * rxf[2];
  row[4] = R[1] * rxf[0] + R[4] * rxf[1] + R[7] * rxf[2];
  row[5] = R[2] * rxf[0] + R[5] * rxf[1] + R[8] * rxf[2];
}

// elliptic cone (condim 3, friction mu_geom on both tangents): force and Hessian block at row residuals jar.
// Returns the cost.  Zones: top (free), bottom (quadratic), middle.
D3IL_HD double cone_eval(const double* jar, double Dn, double Dt, double mu, double fric, double* force, double* Hc /* 3x3 */) {
  if (Dn == 0) {   // inert row (inactive contact slot of this lane inside a wave-uniform loop)
#pragma unroll
    for (int i = 0; i < 9; i++) Hc[i] = 0;
    force[0] = force[1] = force[2] = 0;
    return 0;
  }
  double U0 = jar[0] * mu, U1 = jar[1] * fric, U2 = jar[2] * fric;
  double T2 = U1 * U1 + U2 * U2;
  double iT = T2 > 0 ? rsqrtd(T2) : 0.0;
  double N = U0, T = T2 * iT;
#pragma unroll
  for (int i = 0; i < 9; i++) Hc[i] = 0;
  if (N >= mu * T || (T <= 0 && N >= 0)) { force[0] = force[1] = force[2] = 0; return 0; }
  if (mu * N + T <= 0 || (T <= 0 && N < 0)) {
    force[0] = -Dn * jar[0]; force[1] = -Dt * jar[1]; force[2] = -Dt * jar[2];
    Hc[0] = Dn; Hc[4] = Dt; Hc[8] = Dt;
    return 0.5 * (Dn * jar[0] * jar[0] + Dt * jar[1] * jar[1] + Dt * jar[2] * jar[2]);
  }
  double Dm = Dn * rcpd(fmax(1e-15, mu * mu * (1 + mu * mu))), NmT = N - mu * T;
  double iT3 = iT * iT * iT;
  double g[3] = {mu, -mu * fric * U1 * iT, -mu * fric * U2 * iT}, U[3] = {0, U1, U2};
#pragma unroll
  for (int j = 0; j < 3; j++) force[j] = -Dm * NmT * g[j];
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int b = 0; b < 3; b++) {
      double h = g[a] * g[b];
      if (a > 0 && b > 0) h += NmT * (-mu) * fric * fric * ((a == b ? iT : 0) - U[a] * U[b] * iT3);
      Hc[3 * a + b] = Dm * h;
    }
  return 0.5 * Dm * NmT * NmT;
}

// wave-level OR of a per-lane predicate (host: identity)
D3IL_HD bool wave_any(bool p) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __any(p) != 0;
#else
  return p;
#endif
}

// ------------------------------------------------------------------------------------------------ cube <-> table slabs
// box_box() specialised to a cube on one of the two large axis-aligned static slabs, away from the slab edges: the slab top
// is the reference face (see the separating-axis argument in DESIGN.md), the incident face is the cube face most opposed to
// +z, nothing is clipped, and each of its four vertices below the slab top is a contact (normal +z, frame z / y / -x,
// position midway).  Slot i = 4 * slab + vertex.  Same numbers as box_box() in this regime (tests/test_push_kernel_host.py).
struct CubeSlab {
  double r[8][3];     // contact position relative to the cube centre
  double dist[8];     // signed distance (< 0: active)
};
struct CubeFace { double Bk[3], B1[3], B2[3], hk, h1, h2, sgi; };
D3IL_HD void cube_face(const PushConsts& pc_, const double* R, CubeFace& f) {   // incident face: the cube face most opposed to +z
  D3IL_PUSH_CONSTS(pc_, pc);
  double bz0 = fabs(R[6]), bz1 = fabs(R[7]), bz2 = fabs(R[8]);
  int kin = 0; double bestdot = bz0;
  if (bz1 > bestdot) { bestdot = bz1; kin = 1; }
  if (bz2 > bestdot) { bestdot = bz2; kin = 2; }
#pragma unroll
  for (int k = 0; k < 3; k++) {
    double c0 = R[3 * k], c1 = R[3 * k + 1], c2 = R[3 * k + 2];
    f.Bk[k] = kin == 0 ? c0 : (kin == 1 ? c1 : c2);
    f.B1[k] = kin == 0 ? c1 : (kin == 1 ? c2 : c0);
    f.B2[k] = kin == 0 ? c2 : (kin == 1 ? c0 : c1);
  }
  f.hk = kin == 0 ? pc.box_half[0] : (kin == 1 ? pc.box_half[1] : pc.box_half[2]);
  f.h1 = kin == 0 ? pc.box_half[1] : (kin == 1 ? pc.box_half[2] : pc.box_half[0]);
  f.h2 = kin == 0 ? pc.box_half[2] : (kin == 1 ? pc.box_half[0] : pc.box_half[1]);
  f.sgi = f.Bk[2] > 0 ? -1.0 : 1.0;
}
// slot i = 4 * slab + vertex: contact position relative to the cube centre and signed distance
D3IL_HD void slot_geom(const PushConsts& pc_, const CubeFace& f, const double* pos, int i, double* r, double* dist) {
  D3IL_PUSH_CONSTS(pc_, pc);
  int s = i >> 2, v = i & 3;
  double c0 = (v == 0 || v == 3) ? 1.0 : -1.0, c1 = v < 2 ? 1.0 : -1.0, x[3];
  double sc0 = s ? pc.slab_c[1][0] : pc.slab_c[0][0], sc1 = s ? pc.slab_c[1][1] : pc.slab_c[0][1], sc2 = s ? pc.slab_c[1][2] : pc.slab_c[0][2];
  double sh2 = s ? pc.slab_h[1][2] : pc.slab_h[0][2];
  const double scv[3] = {sc0, sc1, sc2};
#pragma unroll
  for (int k = 0; k < 3; k++) x[k] = pos[k] + f.sgi * f.hk * f.Bk[k] + c0 * f.h1 * f.B1[k] + c1 * f.h2 * f.B2[k] - scv[k];
  double w = x[2] - sh2;
  *dist = w;
  r[0] = sc0 + x[0] - pos[0];
  r[1] = sc1 + x[1] - pos[1];
  r[2] = sc2 + (sh2 + 0.5 * w) - pos[2];
}
D3IL_HD void cube_slab_contacts(const PushConsts& pc_, const double* pos, const double* R, CubeSlab& cs) {
  D3IL_PUSH_CONSTS(pc_, pc);
  CubeFace f;
  cube_face(pc, R, f);
#pragma unroll
  for (int i = 0; i < 8; i++) slot_geom(pc, f, pos, i, cs.r[i], &cs.dist[i]);
}
// rows of slot i: normal z, tangents y and -x (make_frame of (0, 0, 1))
D3IL_HD void slab_rows(const double* R, const double* r, double (*J)[6]) {
  const double fz[3] = {0, 0, 1}, fy[3] = {0, 1, 0}, fx[3] = {-1, 0, 0};
  box_row_r(R, r, fz, J[0]); box_row_r(R, r, fy, J[1]); box_row_r(R, r, fx, J[2]);
}

// One free cube resting / sliding on the slabs, no other contact: 6-dof primal Newton with exact line search, everything
// in registers.  x: start point in, optimum out (acceleration: linear world, angular body axes).
D3IL_NOINLINE inline bool cube_newton(const PushConsts& pc_, const double* R, const double* vel, const CubeSlab& cs, const double* a0, double* x) {
  D3IL_PUSH_CONSTS(pc_, pc);
  const double Mm[6] = {pc.box_mass, pc.box_mass, pc.box_mass, pc.box_inertia, pc.box_inertia, pc.box_inertia};
  const double fric = pc.ct_fric[0], mu = fric * sqrt(1 / fmax(1e-15, pc.impratio)), impr = pc.impratio;
  bool act[8]; double aref[8][3], Dn[8];
  bool any = false;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    act[i] = cs.dist[i] < 0;
    any = any || act[i];
    aref[i][0] = aref[i][1] = aref[i][2] = 0; Dn[i] = 0;
  }
  if (!any) {
#pragma unroll
    for (int k = 0; k < 6; k++) x[k] = a0[k];
    return true;
  }
  bool converged = false;
  D3IL_STAT(g_stats.newton_calls++);
  for (int it = 0; it < PUSH_MAXIT && !converged; it++) {
    D3IL_STAT(g_stats.eig_calls++);
    double g[6], H[21], jar[8][3];
#pragma unroll
    for (int k = 0; k < 6; k++) g[k] = Mm[k] * (x[k] - a0[k]);
#pragma unroll
    for (int i = 0; i < 21; i++) H[i] = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) H[tri(k, k)] = Mm[k];
#pragma unroll
    for (int i = 0; i < 8; i++) {
      jar[i][0] = jar[i][1] = jar[i][2] = 0;
      if (wave_any(act[i])) {
        double J[3][6]; slab_rows(R, cs.r[i], J);
        if (it == 0 && act[i]) {   // reference acceleration and regularisation of the contact, with the rows that are needed anyway
          double imp = impedance(pc.ct_solimp[0], cs.dist[i]);
          double v3[3];
#pragma unroll
          for (int r = 0; r < 3; r++) { double a = 0;
#pragma unroll
            for (int k = 0; k < 6; k++) a += J[r][k] * vel[k]; v3[r] = a; }
          aref[i][0] = -pc.ct_B[0] * v3[0] - pc.ct_K[0] * imp * cs.dist[i];
          aref[i][1] = -pc.ct_B[0] * v3[1]; aref[i][2] = -pc.ct_B[0] * v3[2];
          Dn[i] = 1 / fmax(1e-15, (1 - imp) / imp * pc.box_invw_t);
        }
        double f[3], Hc[9], jr[3];
#pragma unroll
        for (int r = 0; r < 3; r++) { double a = -aref[i][r];
#pragma unroll
          for (int k = 0; k < 6; k++) a += J[r][k] * x[k]; jr[r] = a; jar[i][r] = a; }
        cone_eval(jr, Dn[i], Dn[i] * impr, mu, fric, f, Hc);
        if (act[i]) {
#pragma unroll
          for (int k = 0; k < 6; k++) g[k] -= J[0][k] * f[0] + J[1][k] * f[1] + J[2][k] * f[2];
#pragma unroll
          for (int a = 0; a < 6; a++) {
            double ta[3];
#pragma unroll
            for (int r = 0; r < 3; r++) ta[r] = Hc[3 * r] * J[0][a] + Hc[3 * r + 1] * J[1][a] + Hc[3 * r + 2] * J[2][a];
#pragma unroll
            for (int b = 0; b <= a; b++) H[tri(a, b)] += ta[0] * J[0][b] + ta[1] * J[1][b] + ta[2] * J[2][b];
          }
        }
      }
    }
    {   // MuJoCo's Newton stops on a scaled gradient below `tolerance` (1e-10); here: absolute, in N and N m
      double gm = 0;
#pragma unroll
      for (int k = 0; k < 6; k++) gm = fmax(gm, fabs(g[k]));
      if (gm <= PUSH_GRAD_TOL) { converged = true; break; }
    }
    double L[21], d[6], id[6], p[6], ng[6]; int nneg;
#pragma unroll
    for (int i = 0; i < 21; i++) L[i] = 0;
    if (!ldl6(H, 0.0, L, d, id, &nneg) || nneg) return false;
#pragma unroll
    for (int k = 0; k < 6; k++) ng[k] = -g[k];
    ldl6_solve(L, id, ng, p);
    double pMp = 0, pMa = 0, jp[8][3];
#pragma unroll
    for (int k = 0; k < 6; k++) { pMp += Mm[k] * p[k] * p[k]; pMa += Mm[k] * p[k] * (x[k] - a0[k]); }
#pragma unroll
    for (int i = 0; i < 8; i++) {
      jp[i][0] = jp[i][1] = jp[i][2] = 0;
      if (wave_any(act[i])) {
        double J[3][6]; slab_rows(R, cs.r[i], J);
#pragma unroll
        for (int r = 0; r < 3; r++) { double a = 0;
#pragma unroll
          for (int k = 0; k < 6; k++) a += J[r][k] * p[k]; jp[i][r] = a; }
      }
    }
    double gTp = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) gTp += g[k] * p[k];
    double alpha = 1, lo = 0, hi = -1, best = 1, wprev = 1e300;
    for (int ls = 0; ls < 40; ls++) {
      D3IL_STAT(g_stats.ik_calls++);
      double d1 = pMa + alpha * pMp, d2 = pMp;
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if (wave_any(act[i])) {
          double jt[3] = {jar[i][0] + alpha * jp[i][0], jar[i][1] + alpha * jp[i][1], jar[i][2] + alpha * jp[i][2]}, ft[3], Hc[9];
          cone_eval(jt, Dn[i], Dn[i] * impr, mu, fric, ft, Hc);
          if (act[i]) {
#pragma unroll
            for (int r = 0; r < 3; r++) { d1 -= ft[r] * jp[i][r];
#pragma unroll
              for (int q = 0; q < 3; q++) d2 += jp[i][r] * Hc[3 * r + q] * jp[i][q]; }
          }
        }
      }
      best = alpha;
      // full Newton step: accepted on the curvature condition phi'(1) <= 0.1 |phi'(0)| (still descending, or just past the minimum)
      if (ls == 0 && d1 <= D3IL_TOL.ls_full * fabs(gTp)) break;
      if (fabs(d1) <= D3IL_TOL.ls_c2 * fabs(gTp) || fabs(d1) <= D3IL_TOL.ls_rel * d2 * alpha || fabs(d1) < 1e-14 * fmax(1.0, fabs(pMa))) break;   // minimiser of phi within 0.1 % of alpha (any such step keeps Newton's rate), or slope at round-off level
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
    double smax = 0, xmax = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) { double dxk = best * p[k]; x[k] += dxk; smax = fmax(smax, fabs(dxk)); xmax = fmax(xmax, fabs(x[k])); }
    // Newton converges quadratically once the full step is accepted: a step below 1e-6 (relative) leaves an error of the
    // order of its square, so the confirming iteration is skipped
    if (smax <= 1e-12 * (1 + xmax) || (best == 1.0 && smax <= D3IL_TOL.step_rel * (1 + xmax))) converged = true;
  }
  return converged;
}

// ------------------------------------------------------------------------------------------------ general path (memory resident)
// Used when the rod touches a cube, the cubes touch each other, or an arm joint is at a limit: all 21 dofs in one Newton
// solve.  Data lives in the scratch areas (contact records, M, a0, x, direction in HBM; Hessian in LDS), the function is
// out of line and keeps few registers live.  g-area inputs: PG_M, PG_A0, contact records (pos, normal, dist, kind, cube),
// PG_JA, PG_AUX (cube poses, velocities, limit rows); PG_X holds the start point and receives the optimum.
constexpr int PG_AUX = PG_GRAD + PUSH_NV;      // R[2][9] pos[2][3] vel21 lim[9][3]
constexpr int PG_AUX_R = PG_AUX, PG_AUX_POS = PG_AUX + 18, PG_AUX_VEL = PG_AUX + 24, PG_AUX_LIM = PG_AUX + 45;

struct SRow { int o1, o2, n2; double v1[6], v2[7]; };   // block 1: 6 cube dofs at o1; block 2: n2 (0, 6 or 7) dofs at o2

D3IL_HD void contact_rows(const PushScratch sc, int ci, SRow* rows) {
  int base = PG_CON + ci * PREC;
  double p[3] = {PGS(base), PGS(base + 1), PGS(base + 2)};
  int kind = (int)PGS(base + 13), cube = (int)PGS(base + 14);
  double Rc[9], r[3], R1[9], r1[3];
  int cb = kind == CK_BOXBOX ? 0 : cube;
#pragma unroll
  for (int k = 0; k < 9; k++) Rc[k] = PGS(PG_AUX_R + 9 * cb + k);
#pragma unroll
  for (int k = 0; k < 3; k++) r[k] = p[k] - PGS(PG_AUX_POS + 3 * cb + k);
  if (kind == CK_BOXBOX) {
#pragma unroll
    for (int k = 0; k < 9; k++) R1[k] = PGS(PG_AUX_R + 9 + k);
#pragma unroll
    for (int k = 0; k < 3; k++) r1[k] = p[k] - PGS(PG_AUX_POS + 3 + k);
  }
#pragma unroll
  for (int rr = 0; rr < 3; rr++) {
    double f[3] = {PGS(base + 3 + 3 * rr), PGS(base + 4 + 3 * rr), PGS(base + 5 + 3 * rr)};
    SRow& s = rows[rr];
    box_row_r(Rc, r, f, s.v1);
#pragma unroll
    for (int k = 0; k < 7; k++) s.v2[k] = 0;
    if (kind == CK_SLAB) { s.o1 = 6 * cube; s.o2 = 0; s.n2 = 0; }
    else if (kind == CK_BOXBOX) {
      s.o1 = 0; s.o2 = 6; s.n2 = 6;
#pragma unroll
      for (int k = 0; k < 6; k++) s.v1[k] = -s.v1[k];
      double t[6]; box_row_r(R1, r1, f, t);
#pragma unroll
      for (int k = 0; k < 6; k++) s.v2[k] = t[k];
    } else {
      s.o1 = 6 * cube; s.o2 = PUSH_ARM0; s.n2 = 7;
#pragma unroll
      for (int k = 0; k < 6; k++) s.v1[k] = -s.v1[k];
#pragma unroll
      for (int k = 0; k < 7; k++) s.v2[k] = PGS(PG_JA + cube * 21 + rr * 7 + k);
    }
  }
}
// row . vector stored in the g area at offset `vec`
D3IL_HD double srow_dot_g(const PushScratch sc, const SRow& s, int vec) {
  double a = 0;
#pragma unroll
  for (int k = 0; k < 6; k++) a += s.v1[k] * PGS(vec + s.o1 + k);
#pragma unroll
  for (int k = 0; k < 7; k++) if (k < s.n2) a += s.v2[k] * PGS(vec + s.o2 + k);
  return a;
}

// skyline Cholesky in the h area.  first[i] = first column of row i inside the envelope: cube1 rows 0, cube2 rows
// (bb ? 0 : 6), arm rows (rod on cube1 ? 0 : rod on cube2 ? 6 : 12)
D3IL_HD int sky_first(int i, bool bb, bool rod1, bool rod2) {
  if (i < 6) return 0;
  if (i < 12) return bb ? 0 : 6;
  return rod1 ? 0 : (rod2 ? 6 : 12);
}
D3IL_HD bool sky_chol(const PushScratch sc, bool bb, bool rod1, bool rod2) {
  bool ok = true;
  for (int i = 0; i < PUSH_NV; i++) {
    int fi = sky_first(i, bb, rod1, rod2);
    for (int j = fi; j <= i; j++) {
      int fj = sky_first(j, bb, rod1, rod2), k0 = fi > fj ? fi : fj;
      double s = PHS(tri(i, j));
      for (int k = k0; k < j; k++) s -= PHS(tri(i, k)) * PHS(tri(j, k));
      if (i == j) { if (!(s > 0)) { ok = false; s = 1; } PHS(tri(i, i)) = sqrt(s); }
      else PHS(tri(i, j)) = s / PHS(tri(j, j));
    }
  }
  return ok;
}
// solves in place on the vector at g offset `vec`
D3IL_HD void sky_solve_g(const PushScratch sc, bool bb, bool rod1, bool rod2, int vec) {
  for (int i = 0; i < PUSH_NV; i++) {
    int fi = sky_first(i, bb, rod1, rod2);
    double s = PGS(vec + i);
    for (int k = fi; k < i; k++) s -= PHS(tri(i, k)) * PGS(vec + k);
    PGS(vec + i) = s / PHS(tri(i, i));
  }
  for (int i = PUSH_NV - 1; i >= 0; i--) {
    double xi = PGS(vec + i) / PHS(tri(i, i));
    PGS(vec + i) = xi;
    int fi = sky_first(i, bb, rod1, rod2);
    for (int k = fi; k < i; k++) PGS(vec + k) -= PHS(tri(i, k)) * xi;
  }
}
D3IL_HD double gM(const PushConsts& pc_, const PushScratch sc, int i, int k) {   // entry (i, k) of the block-diagonal mass matrix
  D3IL_PUSH_CONSTS(pc_, pc);
  if (i < PUSH_ARM0 || k < PUSH_ARM0) return i == k ? ((i % 6) < 3 ? pc.box_mass : pc.box_inertia) : 0.0;
  int a = i - PUSH_ARM0, b = k - PUSH_ARM0;
  return PGS(PG_M + (a >= b ? tri(a, b) : tri(b, a)));
}

// returns false if the solver did not converge
D3IL_NOINLINE inline bool push_general_solve(const PushConsts& pc_, const PushScratch sc, int ncon, bool env_bb, bool env_r1, bool env_r2) {
  D3IL_PUSH_CONSTS(pc_, pc);
  const double impr = pc.impratio;
  // per-contact frame, reference acceleration, regularisation
  for (int ci = 0; ci < ncon; ci++) {
    int base = PG_CON + ci * PREC;
    double n[3] = {PGS(base + 3), PGS(base + 4), PGS(base + 5)}, t1[3], t2[3];
    make_frame(n, t1, t2);
    for (int k = 0; k < 3; k++) { PGS(base + 6 + k) = t1[k]; PGS(base + 9 + k) = t2[k]; }
    int kind = (int)PGS(base + 13), set = kind == CK_SLAB ? 0 : 1;
    double dist = PGS(base + 12);
    double imp = impedance(pc.ct_solimp[set], dist);
    double invw = kind == CK_SLAB ? pc.box_invw_t : (kind == CK_BOXBOX ? 2 * pc.box_invw_t : pc.box_invw_t + PGS(PG_AUX_LIM + 27));
    double Rn = fmax(1e-15, (1 - imp) / imp * invw);
    SRow rows[3];
    contact_rows(sc, ci, rows);
    double v0 = srow_dot_g(sc, rows[0], PG_AUX_VEL), v1 = srow_dot_g(sc, rows[1], PG_AUX_VEL), v2 = srow_dot_g(sc, rows[2], PG_AUX_VEL);
    PGS(base + 15) = -pc.ct_B[set] * v0 - pc.ct_K[set] * imp * dist;
    PGS(base + 16) = -pc.ct_B[set] * v1; PGS(base + 17) = -pc.ct_B[set] * v2;
    PGS(base + 18) = 1 / Rn;
    PGS(base + 19) = pc.ct_fric[set] * sqrt(1 / fmax(1e-15, impr));
  }
  bool converged = false;
  for (int it = 0; it < PUSH_MAXIT && !converged; it++) {
    // gradient (into PG_GRAD) and Hessian (h area) at x
    for (int i = 0; i < PUSH_NV; i++) {
      double s = 0;
      if (i < PUSH_ARM0) s = gM(pc, sc, i, i) * (PGS(PG_X + i) - PGS(PG_A0 + i));
      else for (int k = PUSH_ARM0; k < PUSH_NV; k++) s += gM(pc, sc, i, k) * (PGS(PG_X + k) - PGS(PG_A0 + k));
      PGS(PG_GRAD + i) = s;
    }
    for (int i = 0; i < PUSH_NH; i++) PHS(i) = 0;
    for (int i = 0; i < PUSH_ARM0; i++) PHS(tri(i, i)) = gM(pc, sc, i, i);
    for (int i = 0; i < NDOF; i++) for (int k = 0; k <= i; k++) PHS(tri(PUSH_ARM0 + i, PUSH_ARM0 + k)) = PGS(PG_M + tri(i, k));
    for (int k = 0; k < NDOF; k++) {
      double sign = PGS(PG_AUX_LIM + 3 * k), D = PGS(PG_AUX_LIM + 3 * k + 1), aref = PGS(PG_AUX_LIM + 3 * k + 2);
      if (sign != 0) {
        double jar = sign * PGS(PG_X + PUSH_ARM0 + k) - aref;
        if (jar < 0) { PGS(PG_GRAD + PUSH_ARM0 + k) += sign * D * jar; PHS(tri(PUSH_ARM0 + k, PUSH_ARM0 + k)) += D; }
      }
    }
    for (int ci = 0; ci < ncon; ci++) {
      int base = PG_CON + ci * PREC;
      SRow rows[3];
      contact_rows(sc, ci, rows);
      double jar[3], force[3], Hc[9];
#pragma unroll
      for (int r = 0; r < 3; r++) { jar[r] = srow_dot_g(sc, rows[r], PG_X) - PGS(base + 15 + r); PGS(base + 20 + r) = jar[r]; }
      double Dn = PGS(base + 18), mu = PGS(base + 19), fric = pc.ct_fric[(int)PGS(base + 13) == CK_SLAB ? 0 : 1];
      cone_eval(jar, Dn, Dn * impr, mu, fric, force, Hc);
      if (force[0] == 0 && force[1] == 0 && force[2] == 0) continue;
      const int o1 = rows[0].o1, o2 = rows[0].o2, n2 = rows[0].n2;
#pragma unroll
      for (int k = 0; k < 6; k++) PGS(PG_GRAD + o1 + k) -= rows[0].v1[k] * force[0] + rows[1].v1[k] * force[1] + rows[2].v1[k] * force[2];
#pragma unroll
      for (int k = 0; k < 7; k++) if (k < n2) PGS(PG_GRAD + o2 + k) -= rows[0].v2[k] * force[0] + rows[1].v2[k] * force[1] + rows[2].v2[k] * force[2];
      // H += J' Hc J : block (1,1), then (2,1) and (2,2)
#pragma unroll
      for (int a = 0; a < 6; a++) {
        double ta[3];
#pragma unroll
        for (int r = 0; r < 3; r++) ta[r] = Hc[3 * r] * rows[0].v1[a] + Hc[3 * r + 1] * rows[1].v1[a] + Hc[3 * r + 2] * rows[2].v1[a];
#pragma unroll
        for (int b = 0; b <= a; b++) PHS(tri(o1 + a, o1 + b)) += ta[0] * rows[0].v1[b] + ta[1] * rows[1].v1[b] + ta[2] * rows[2].v1[b];
      }
#pragma unroll
      for (int a = 0; a < 7; a++) if (a < n2) {
        double ta[3];
#pragma unroll
        for (int r = 0; r < 3; r++) ta[r] = Hc[3 * r] * rows[0].v2[a] + Hc[3 * r + 1] * rows[1].v2[a] + Hc[3 * r + 2] * rows[2].v2[a];
#pragma unroll
        for (int b = 0; b < 6; b++) PHS(tri(o2 + a, o1 + b)) += ta[0] * rows[0].v1[b] + ta[1] * rows[1].v1[b] + ta[2] * rows[2].v1[b];
#pragma unroll
        for (int b = 0; b < 7; b++) if (b <= a) PHS(tri(o2 + a, o2 + b)) += ta[0] * rows[0].v2[b] + ta[1] * rows[1].v2[b] + ta[2] * rows[2].v2[b];
      }
    }
    if (!sky_chol(sc, env_bb, env_r1, env_r2)) return false;
    // direction p = -H^-1 grad, in place
    for (int k = 0; k < PUSH_NV; k++) PGS(PG_GRAD + k) = -PGS(PG_GRAD + k);
    sky_solve_g(sc, env_bb, env_r1, env_r2, PG_GRAD);
    double pMp = 0, pMa = 0;
    for (int i = 0; i < PUSH_NV; i++) {
      double s = 0, sa = 0;
      if (i < PUSH_ARM0) { double mm = gM(pc, sc, i, i); s = mm * PGS(PG_GRAD + i); sa = mm * (PGS(PG_X + i) - PGS(PG_A0 + i)); }
      else for (int k = PUSH_ARM0; k < PUSH_NV; k++) { double mm = gM(pc, sc, i, k); s += mm * PGS(PG_GRAD + k); sa += mm * (PGS(PG_X + k) - PGS(PG_A0 + k)); }
      pMp += PGS(PG_GRAD + i) * s; pMa += PGS(PG_GRAD + i) * sa;
    }
    for (int ci = 0; ci < ncon; ci++) {
      int base = PG_CON + ci * PREC;
      SRow rows[3];
      contact_rows(sc, ci, rows);
#pragma unroll
      for (int r = 0; r < 3; r++) PGS(base + 23 + r) = srow_dot_g(sc, rows[r], PG_GRAD);
    }
    double alpha = 1, lo = 0, hi = -1, best = 1, wprev = 1e300;
    for (int ls = 0; ls < 40; ls++) {
      double d1 = pMa + alpha * pMp, d2 = pMp;
      for (int k = 0; k < NDOF; k++) {
        double sign = PGS(PG_AUX_LIM + 3 * k), D = PGS(PG_AUX_LIM + 3 * k + 1), aref = PGS(PG_AUX_LIM + 3 * k + 2);
        if (sign != 0) {
          double jp = sign * PGS(PG_GRAD + PUSH_ARM0 + k), jar = sign * PGS(PG_X + PUSH_ARM0 + k) - aref + alpha * jp;
          if (jar < 0) { d1 += D * jar * jp; d2 += D * jp * jp; }
        }
      }
      for (int ci = 0; ci < ncon; ci++) {
        int base = PG_CON + ci * PREC;
        double jp[3] = {PGS(base + 23), PGS(base + 24), PGS(base + 25)};
        double jt[3] = {PGS(base + 20) + alpha * jp[0], PGS(base + 21) + alpha * jp[1], PGS(base + 22) + alpha * jp[2]}, ft[3], Hc[9];
        double Dn = PGS(base + 18), mu = PGS(base + 19), fric = pc.ct_fric[(int)PGS(base + 13) == CK_SLAB ? 0 : 1];
        cone_eval(jt, Dn, Dn * impr, mu, fric, ft, Hc);
#pragma unroll
        for (int r = 0; r < 3; r++) { d1 -= ft[r] * jp[r];
#pragma unroll
          for (int q = 0; q < 3; q++) d2 += jp[r] * Hc[3 * r + q] * jp[q]; }
      }
      best = alpha;
      if (fabs(d1) <= D3IL_TOL.ls_rel * d2 * alpha || fabs(d1) < 1e-14 * fmax(1.0, fabs(pMa))) break;   // minimiser of phi within 0.1 % of alpha (any such step keeps Newton's rate), or slope at round-off level
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
    double smax = 0, xmax = 0;
    for (int k = 0; k < PUSH_NV; k++) {
      double dxk = best * PGS(PG_GRAD + k), xn = PGS(PG_X + k) + dxk;
      PGS(PG_X + k) = xn; smax = fmax(smax, fabs(dxk)); xmax = fmax(xmax, fabs(xn));
    }
    // Newton converges quadratically once the full step is accepted: a step below 1e-6 (relative) leaves an error of the
    // order of its square, so the confirming iteration is skipped
    if (smax <= 1e-12 * (1 + xmax) || (best == 1.0 && smax <= D3IL_TOL.step_rel * (1 + xmax))) converged = true;
  }
  return converged;
}

// ------------------------------------------------------------------------------------------------ coupled path (registers + LDS)
// Rod on one cube and / or cube <-> cube contacts: Newton over arm + both cubes with the arm eliminated first.  The rod
// contact row is r = A a_arm + Jb a_cube - aref (A: 3 x 7 arm Jacobian, Jb: minus the cube rows), so with Hc the 3 x 3 cone
// Hessian:  H_aa = M_a + limits + A' Hc A,  C = A' Hc Jb,  and the arm drops out of the Newton system as
//   (H_cc - Jb' K Jb) p_c = -g_c + Jb' Hc (A z),   K = Hc (A Z) Hc,   Z = H_aa^-1 A',  z = H_aa^-1 g_a,
//   p_a = -z - Z Hc (Jb p_c).
// H_cc is the 12 x 12 cube block (cube-cube contacts fill its off-diagonal 6 x 6), factorised unrolled in registers; the
// 9 x 9 arm factorisation reuses ldl9.  Per-contact data (aref, D, row residuals jar, directional derivatives jp) sits
// in the lane-strided LDS table.  Layout of the table (fields per lane):
constexpr int PT_X = 0, PT_P = 21, PT_A0 = 42, PT_M = 63, PT_LIM = 108, PT_R = 135, PT_POS = 153, PT_VEL = 159, PT_JA = 180, PT_Z = 201, PT_ZG = 228;
constexpr int PT_SLAB = 237;            // 16 slots x (aref[3], Dn, jar[3], jp[3])
constexpr int PT_CON = PT_SLAB + 160;   // 8 cube-cube + 1 rod: pos[3] n[3] dist aref[3] Dn jar[3] jp[3]
constexpr int PT_ROD = PT_CON + 8 * 17;
constexpr int PT_G = PT_VEL;                // gradient of the current iterate (the velocities are only needed by the set-up pass)
constexpr int PT_H = PT_CON + 9 * 17;      // 12 x 12 cube Hessian, packed lower (78)
constexpr int PT_XR = PT_H + 78;           // rod-contact hand-over between the phases (42)
constexpr int PT_PAIR = PT_XR + 42;        // lane-pair hand-over: per cube pos[3] quat[4] vel[6] warm[6] (2 x 19), then SOLVED, PFLAG
constexpr int PT_SOLVED = PT_PAIR + 38, PT_PFLAG = PT_PAIR + 39;
constexpr int PT_SIZE = PT_PAIR + 40;      // 710

template <int N> D3IL_HD bool ldl_n(double* A, double* d, double* id) {   // in place: strict lower part of A becomes L
  bool ok = true;
#pragma unroll
  for (int j = 0; j < N; j++) {
    double s = A[tri(j, j)];
#pragma unroll
    for (int k = 0; k < j; k++) s -= A[tri(j, k)] * A[tri(j, k)] * d[k];
    if (!(s > 1e-300)) { s = 1; ok = false; }
    d[j] = s;
    double inv = rcpd(s);
    id[j] = inv;
#pragma unroll
    for (int i = j + 1; i < N; i++) {
      double t = A[tri(i, j)];
#pragma unroll
      for (int k = 0; k < j; k++) t -= A[tri(i, k)] * A[tri(j, k)] * d[k];
      A[tri(i, j)] = t * inv;
    }
  }
  return ok;
}
template <int N> D3IL_HD void ldl_solve_n(const double* L, const double* id, double* x) {
#pragma unroll
  for (int i = 0; i < N; i++) {
#pragma unroll
    for (int k = 0; k < i; k++) x[i] -= L[tri(i, k)] * x[k];
  }
#pragma unroll
  for (int i = 0; i < N; i++) x[i] *= id[i];
#pragma unroll
  for (int i = N - 1; i >= 0; i--) {
#pragma unroll
    for (int k = i + 1; k < N; k++) x[i] -= L[tri(k, i)] * x[k];
  }
}
// accumulate J' Hc J into the packed lower-triangular 12 x 12 block: rows/cols at offsets oa >= ob (6 wide each)
D3IL_HD void acc_block(double* H, int oa, int ob, const double (*Ja)[6], const double (*Jb)[6], const double* Hc, bool diag) {
#pragma unroll
  for (int a = 0; a < 6; a++) {
    double ta[3];
#pragma unroll
    for (int r = 0; r < 3; r++) ta[r] = Hc[3 * r] * Ja[0][a] + Hc[3 * r + 1] * Ja[1][a] + Hc[3 * r + 2] * Ja[2][a];
#pragma unroll
    for (int b = 0; b < 6; b++) if (!diag || b <= a) H[tri(oa + a, ob + b)] += ta[0] * Jb[0][b] + ta[1] * Jb[1][b] + ta[2] * Jb[2][b];
  }
}

// nbb: cube-cube contacts in the table; rod_cube: cube group touched by the rod (-1: none).  x: PT_X in / out.
// two (wave-uniform): both cube groups are in the system.  When no lane of the wave has a cube-cube contact the caller puts
// the cube under the rod into group 0 and only that group is processed (6 x 6 cube system, half the slab slots); the other
// cube then goes through its decoupled solve.
// The function is written as a sequence of phases that hand their results over through the LDS table, so that the live
// register set of each phase stays small (the 12 x 12 Hessian is only in registers while it is factorised).
D3IL_NOINLINE inline bool coupled_newton(const PushConsts& pc_, const PushScratch sc, int nbb, int rod_cube, double rod_invw, bool two) {
  D3IL_PUSH_CONSTS(pc_, pc);
  const int nb = two ? PUSH_NB : 1, ncd = 6 * nb;
  // This lane's own system holds group 1 only with a cube-cube contact.  In a wave that processes both groups for the sake of
  // another lane, group 1 of this lane is inert (no contacts, x = a0 => zero gradient, zero step) and is left out of the
  // convergence measures, so the result does not depend on which environments share a wave.
  const bool lane_two = nbb > 0;
  const double impr = pc.impratio, isq = sqrt(1 / fmax(1e-15, impr));
  const double fric0 = pc.ct_fric[0], mu0 = fric0 * isq, fric1 = pc.ct_fric[1], mu1 = fric1 * isq;
  const bool rod = rod_cube >= 0;
  const bool any_rod = wave_any(rod);
  int nbb_max = nbb;
#if defined(__HIP_DEVICE_COMPILE__)
  for (int o = 32; o > 0; o >>= 1) { int t = __shfl_xor(nbb_max, o); nbb_max = t > nbb_max ? t : nbb_max; }
  nbb_max = nbb_max > 8 ? 8 : (nbb_max < nbb ? nbb : nbb_max);   // lanes outside the branch contribute stale registers
#endif
  PUSH_TIC;
  // ---- per-contact reference acceleration and regularisation
#pragma clang loop unroll(disable)
  for (int b = 0; b < nb; b++) {
    double R[9], pos[3], vel[6];
#pragma unroll
    for (int k = 0; k < 9; k++) R[k] = PTS(PT_R + 9 * b + k);
#pragma unroll
    for (int k = 0; k < 3; k++) pos[k] = PTS(PT_POS + 3 * b + k);
#pragma unroll
    for (int k = 0; k < 6; k++) vel[k] = PTS(PT_VEL + 6 * b + k);
    CubeFace face;
    cube_face(pc, R, face);
#pragma clang loop unroll(disable)
    for (int i = 0; i < 8; i++) {
      int base = PT_SLAB + 10 * (8 * b + i);
      double r[3], dist;
      slot_geom(pc, face, pos, i, r, &dist);
      bool act = dist < 0 && (b == 0 || lane_two);
      double ar[3] = {0, 0, 0}, Dn = 0;
      if (wave_any(act)) {
        double J[3][6]; slab_rows(R, r, J);
        double imp = impedance(pc.ct_solimp[0], dist);
        double v3[3];
#pragma unroll
        for (int rr = 0; rr < 3; rr++) { double a = 0;
#pragma unroll
          for (int k = 0; k < 6; k++) a += J[rr][k] * vel[k]; v3[rr] = a; }
        if (act) {
          ar[0] = -pc.ct_B[0] * v3[0] - pc.ct_K[0] * imp * dist; ar[1] = -pc.ct_B[0] * v3[1]; ar[2] = -pc.ct_B[0] * v3[2];
          Dn = 1 / fmax(1e-15, (1 - imp) / imp * pc.box_invw_t);
        }
      }
      PTS(base) = ar[0]; PTS(base + 1) = ar[1]; PTS(base + 2) = ar[2]; PTS(base + 3) = Dn;
    }
  }
  for (int ci = 0; ci < 9; ci++) {
    bool isrod = ci == 8;
    if (isrod ? !any_rod : ci >= nbb_max) continue;
    bool act = isrod ? rod : ci < nbb;
    int base = PT_CON + 17 * ci;
    double ar[3] = {0, 0, 0}, Dn = 0;
    if (act) {
      double p[3] = {PTS(base), PTS(base + 1), PTS(base + 2)}, n[3] = {PTS(base + 3), PTS(base + 4), PTS(base + 5)}, t1[3], t2[3], dist = PTS(base + 6);
      make_frame(n, t1, t2);
      double v3[3] = {0, 0, 0};
      for (int b = 0; b < PUSH_NB; b++) {
        if (isrod && b != rod_cube) continue;
        double R[9], r[3], vel[6], J[3][6];
        for (int k = 0; k < 9; k++) R[k] = PTS(PT_R + 9 * b + k);
        for (int k = 0; k < 3; k++) r[k] = p[k] - PTS(PT_POS + 3 * b + k);
        for (int k = 0; k < 6; k++) vel[k] = PTS(PT_VEL + 6 * b + k);
        box_row_r(R, r, n, J[0]); box_row_r(R, r, t1, J[1]); box_row_r(R, r, t2, J[2]);
        double sg = (isrod || b == 0) ? -1.0 : 1.0;
        for (int r3 = 0; r3 < 3; r3++) for (int k = 0; k < 6; k++) v3[r3] += sg * J[r3][k] * vel[k];
      }
      if (isrod) for (int r3 = 0; r3 < 3; r3++) for (int k = 0; k < NARM; k++) v3[r3] += PTS(PT_JA + 7 * r3 + k) * PTS(PT_VEL + PUSH_ARM0 + k);
      double imp = impedance(pc.ct_solimp[1], dist);
      double invw = isrod ? pc.box_invw_t + rod_invw : 2 * pc.box_invw_t;
      ar[0] = -pc.ct_B[1] * v3[0] - pc.ct_K[1] * imp * dist; ar[1] = -pc.ct_B[1] * v3[1]; ar[2] = -pc.ct_B[1] * v3[2];
      Dn = 1 / fmax(1e-15, (1 - imp) / imp * invw);
    }
    PTS(base + 7) = ar[0]; PTS(base + 8) = ar[1]; PTS(base + 9) = ar[2]; PTS(base + 10) = Dn;
  }
  bool converged = false;
  PUSH_TOC(0);
  D3IL_STAT(g_stats.contact_calls++);
  for (int it = 0; it < PUSH_MAXIT && !converged; it++) {
    D3IL_STAT(g_stats.newton_iters++);
    double gmax_arm = 0;
    // ================= phase A: arm gradient, H_aa, elimination quantities -> PT_ZG, PT_Z, PT_XR
    // PT_XR: W[9] (Hc - K) | hcAz[3] | frod[3] | HcR[9] | JbR[18] (cube rows of the rod contact, negated)
    {
      double xa[NDOF], ga[NDOF], Haa[45];
#pragma unroll
      for (int i = 0; i < 45; i++) Haa[i] = PTS(PT_M + i);
#pragma unroll
      for (int k = 0; k < NDOF; k++) xa[k] = PTS(PT_X + PUSH_ARM0 + k);
      {
        double dx[NDOF];
#pragma unroll
        for (int k = 0; k < NDOF; k++) dx[k] = xa[k] - PTS(PT_A0 + PUSH_ARM0 + k);
        symv9(Haa, dx, ga);
      }
#pragma unroll
      for (int k = 0; k < NDOF; k++) {
        double sign = PTS(PT_LIM + 3 * k), D = PTS(PT_LIM + 3 * k + 1), aref = PTS(PT_LIM + 3 * k + 2);
        double jar = sign * xa[k] - aref;
        if (sign != 0 && jar < 0) { ga[k] += sign * D * jar; Haa[tri(k, k)] += D; }
      }
      double A[3][NARM], HcR[9], frod[3] = {0, 0, 0};
#pragma unroll
      for (int i = 0; i < 9; i++) HcR[i] = 0;
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int k = 0; k < NARM; k++) A[r][k] = 0;
      if (any_rod) {
        double JbR[3][6];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
          for (int k = 0; k < 6; k++) JbR[r][k] = 0;
        if (rod) {
          int base = PT_ROD;
          double p[3] = {PTS(base), PTS(base + 1), PTS(base + 2)}, n[3] = {PTS(base + 3), PTS(base + 4), PTS(base + 5)}, t1[3], t2[3];
          make_frame(n, t1, t2);
          double R[9], r[3];
          int b = rod_cube;
#pragma unroll
          for (int k = 0; k < 9; k++) R[k] = PTS(PT_R + 9 * b + k);
#pragma unroll
          for (int k = 0; k < 3; k++) r[k] = p[k] - PTS(PT_POS + 3 * b + k);
          box_row_r(R, r, n, JbR[0]); box_row_r(R, r, t1, JbR[1]); box_row_r(R, r, t2, JbR[2]);
#pragma unroll
          for (int r3 = 0; r3 < 3; r3++)
#pragma unroll
            for (int k = 0; k < 6; k++) JbR[r3][k] = -JbR[r3][k];
#pragma unroll
          for (int r3 = 0; r3 < 3; r3++)
#pragma unroll
            for (int k = 0; k < NARM; k++) A[r3][k] = PTS(PT_JA + 7 * r3 + k);
          double jar[3];
#pragma unroll
          for (int r3 = 0; r3 < 3; r3++) {
            double a = -PTS(base + 7 + r3);
#pragma unroll
            for (int k = 0; k < NARM; k++) a += A[r3][k] * xa[k];
#pragma unroll
            for (int k = 0; k < 6; k++) a += JbR[r3][k] * PTS(PT_X + 6 * b + k);
            jar[r3] = a; PTS(base + 11 + r3) = a;
          }
          double Dn = PTS(base + 10);
          cone_eval(jar, Dn, Dn * impr, mu1, fric1, frod, HcR);
#pragma unroll
          for (int k = 0; k < NARM; k++) {
            ga[k] -= A[0][k] * frod[0] + A[1][k] * frod[1] + A[2][k] * frod[2];
            double ta[3];
#pragma unroll
            for (int r3 = 0; r3 < 3; r3++) ta[r3] = HcR[3 * r3] * A[0][k] + HcR[3 * r3 + 1] * A[1][k] + HcR[3 * r3 + 2] * A[2][k];
#pragma unroll
            for (int q = 0; q <= k; q++) Haa[tri(k, q)] += ta[0] * A[0][q] + ta[1] * A[1][q] + ta[2] * A[2][q];
          }
        }
#pragma unroll
        for (int i = 0; i < 9; i++) PTS(PT_XR + 15 + i) = HcR[i];
#pragma unroll
        for (int i = 0; i < 3; i++) PTS(PT_XR + 12 + i) = frod[i];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
          for (int k = 0; k < 6; k++) PTS(PT_XR + 24 + 6 * r + k) = JbR[r][k];
      }
      {
        double gm = 0;
#pragma unroll
        for (int k = 0; k < NDOF; k++) { gm = fmax(gm, fabs(ga[k])); PTS(PT_G + PUSH_ARM0 + k) = ga[k]; }
        gmax_arm = gm;
      }
      double da[NDOF], ida[NDOF];
      if (!ldl_n<NDOF>(Haa, da, ida)) return false;
      double z[NDOF];
#pragma unroll
      for (int k = 0; k < NDOF; k++) z[k] = ga[k];
      ldl_solve_n<NDOF>(Haa, ida, z);
#pragma unroll
      for (int k = 0; k < NDOF; k++) PTS(PT_ZG + k) = z[k];
      if (any_rod) {
        double AZ[9], Az[3];
#pragma unroll
        for (int r3 = 0; r3 < 3; r3++) {
          double col[NDOF];
#pragma unroll
          for (int k = 0; k < NDOF; k++) col[k] = k < NARM ? A[r3][k] : 0.0;
          ldl_solve_n<NDOF>(Haa, ida, col);
#pragma unroll
          for (int k = 0; k < NDOF; k++) PTS(PT_Z + 9 * r3 + k) = col[k];
#pragma unroll
          for (int q = 0; q < 3; q++) { double a = 0;
#pragma unroll
            for (int k = 0; k < NARM; k++) a += A[q][k] * col[k]; AZ[3 * q + r3] = a; }
        }
#pragma unroll
        for (int q = 0; q < 3; q++) { double a = 0;
#pragma unroll
          for (int k = 0; k < NARM; k++) a += A[q][k] * z[k]; Az[q] = a; }
        double T[9];
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
          for (int b = 0; b < 3; b++) T[3 * a + b] = HcR[3 * a] * AZ[b] + HcR[3 * a + 1] * AZ[3 + b] + HcR[3 * a + 2] * AZ[6 + b];
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
          for (int b = 0; b < 3; b++) PTS(PT_XR + 3 * a + b) = HcR[3 * a + b] - (T[3 * a] * HcR[b] + T[3 * a + 1] * HcR[3 + b] + T[3 * a + 2] * HcR[6 + b]);
#pragma unroll
        for (int a = 0; a < 3; a++) PTS(PT_XR + 9 + a) = HcR[3 * a] * Az[0] + HcR[3 * a + 1] * Az[1] + HcR[3 * a + 2] * Az[2];
      }
    }
    PUSH_TOC(1);
    // ================= phase B: cube gradients and diagonal Hessian blocks (registers) -> PT_H, PT_P (right-hand side)
    if (two) for (int i = 0; i < 36; i++) PTS(PT_H + tri(6 + i / 6, i % 6)) = 0;      // off-diagonal 6 x 6 block
#pragma clang loop unroll(disable)
    for (int b = 0; b < nb; b++) {
      const double Mc[6] = {pc.box_mass, pc.box_mass, pc.box_mass, pc.box_inertia, pc.box_inertia, pc.box_inertia};
      double R[9], pos[3], xb[6], Hb[21], gb[6];
#pragma unroll
      for (int k = 0; k < 9; k++) R[k] = PTS(PT_R + 9 * b + k);
#pragma unroll
      for (int k = 0; k < 3; k++) pos[k] = PTS(PT_POS + 3 * b + k);
#pragma unroll
      for (int k = 0; k < 6; k++) xb[k] = PTS(PT_X + 6 * b + k);
#pragma unroll
      for (int i = 0; i < 21; i++) Hb[i] = 0;
#pragma unroll
      for (int k = 0; k < 6; k++) { gb[k] = Mc[k] * (xb[k] - PTS(PT_A0 + 6 * b + k)); Hb[tri(k, k)] = Mc[k]; }
      CubeFace face;
      cube_face(pc, R, face);
#pragma clang loop unroll(disable)
      for (int i = 0; i < 8; i++) {
        int base = PT_SLAB + 10 * (8 * b + i);
        double Dn = PTS(base + 3);
        if (wave_any(Dn != 0)) {
          double r[3], dist;
          slot_geom(pc, face, pos, i, r, &dist);
          double J[3][6]; slab_rows(R, r, J);
          double jar[3], f[3], Hc[9];
#pragma unroll
          for (int rr = 0; rr < 3; rr++) { double a = -PTS(base + rr);
#pragma unroll
            for (int k = 0; k < 6; k++) a += J[rr][k] * xb[k]; jar[rr] = a; PTS(base + 4 + rr) = a; }
          cone_eval(jar, Dn, Dn * impr, mu0, fric0, f, Hc);
#pragma unroll
          for (int k = 0; k < 6; k++) gb[k] -= J[0][k] * f[0] + J[1][k] * f[1] + J[2][k] * f[2];
          acc_block(Hb, 0, 0, J, J, Hc, true);
        }
      }
      bool rod_here = false;
      if (any_rod) {
        if (rod_cube == b) {   // rod contact seen from the cube: gradient, reduced Hessian J' W J, reduced right-hand side J' Hc (A z)
          double JbR[3][6], W[9], hf[3];
#pragma unroll
          for (int r = 0; r < 3; r++)
#pragma unroll
            for (int k = 0; k < 6; k++) JbR[r][k] = PTS(PT_XR + 24 + 6 * r + k);
#pragma unroll
          for (int i = 0; i < 9; i++) W[i] = PTS(PT_XR + i);
#pragma unroll
          for (int i = 0; i < 3; i++) hf[i] = PTS(PT_XR + 12 + i);
#pragma unroll
          for (int k = 0; k < 6; k++) gb[k] -= JbR[0][k] * hf[0] + JbR[1][k] * hf[1] + JbR[2][k] * hf[2];      // gradient
#pragma unroll
          for (int k = 0; k < 6; k++) PTS(PT_G + 6 * b + k) = gb[k];
#pragma unroll
          for (int i = 0; i < 3; i++) hf[i] = PTS(PT_XR + 9 + i);
#pragma unroll
          for (int k = 0; k < 6; k++) gb[k] -= JbR[0][k] * hf[0] + JbR[1][k] * hf[1] + JbR[2][k] * hf[2];      // reduced right-hand side
          acc_block(Hb, 0, 0, JbR, JbR, W, true);
          rod_here = true;
        }
      }
      if (!rod_here) {
#pragma unroll
        for (int k = 0; k < 6; k++) PTS(PT_G + 6 * b + k) = gb[k];
      }
#pragma unroll
      for (int a = 0; a < 6; a++)
#pragma unroll
        for (int q = 0; q <= a; q++) PTS(PT_H + tri(6 * b + a, 6 * b + q)) = Hb[tri(a, q)];
#pragma unroll
      for (int k = 0; k < 6; k++) PTS(PT_P + 6 * b + k) = -gb[k];
    }
    // cube <-> cube contacts: read-modify-write on the LDS copy of the Hessian
    for (int ci = 0; ci < nbb_max; ci++) {
      if (ci < nbb) {
        int base = PT_CON + 17 * ci;
        double p[3] = {PTS(base), PTS(base + 1), PTS(base + 2)}, n[3] = {PTS(base + 3), PTS(base + 4), PTS(base + 5)}, t1[3], t2[3];
        make_frame(n, t1, t2);
        double J1[3][6], J2[3][6];
        {
          double R[9], r[3];
#pragma unroll
          for (int k = 0; k < 9; k++) R[k] = PTS(PT_R + k);
#pragma unroll
          for (int k = 0; k < 3; k++) r[k] = p[k] - PTS(PT_POS + k);
          box_row_r(R, r, n, J1[0]); box_row_r(R, r, t1, J1[1]); box_row_r(R, r, t2, J1[2]);
#pragma unroll
          for (int r3 = 0; r3 < 3; r3++)
#pragma unroll
            for (int k = 0; k < 6; k++) J1[r3][k] = -J1[r3][k];
#pragma unroll
          for (int k = 0; k < 9; k++) R[k] = PTS(PT_R + 9 + k);
#pragma unroll
          for (int k = 0; k < 3; k++) r[k] = p[k] - PTS(PT_POS + 3 + k);
          box_row_r(R, r, n, J2[0]); box_row_r(R, r, t1, J2[1]); box_row_r(R, r, t2, J2[2]);
        }
        double jar[3], f[3], Hc[9], Dn = PTS(base + 10);
#pragma unroll
        for (int r3 = 0; r3 < 3; r3++) {
          double a = -PTS(base + 7 + r3);
#pragma unroll
          for (int k = 0; k < 6; k++) a += J1[r3][k] * PTS(PT_X + k) + J2[r3][k] * PTS(PT_X + 6 + k);
          jar[r3] = a; PTS(base + 11 + r3) = a;
        }
        cone_eval(jar, Dn, Dn * impr, mu1, fric1, f, Hc);
#pragma unroll
        for (int k = 0; k < 6; k++) {
          double c1 = J1[0][k] * f[0] + J1[1][k] * f[1] + J1[2][k] * f[2], c2 = J2[0][k] * f[0] + J2[1][k] * f[1] + J2[2][k] * f[2];
          PTS(PT_P + k) += c1; PTS(PT_P + 6 + k) += c2;
          PTS(PT_G + k) -= c1; PTS(PT_G + 6 + k) -= c2;
        }
#pragma unroll
        for (int a = 0; a < 6; a++) {
          double t1a[3], t2a[3];
#pragma unroll
          for (int r = 0; r < 3; r++) {
            t1a[r] = Hc[3 * r] * J1[0][a] + Hc[3 * r + 1] * J1[1][a] + Hc[3 * r + 2] * J1[2][a];
            t2a[r] = Hc[3 * r] * J2[0][a] + Hc[3 * r + 1] * J2[1][a] + Hc[3 * r + 2] * J2[2][a];
          }
#pragma unroll
          for (int q = 0; q < 6; q++) {
            if (q <= a) {
              PTS(PT_H + tri(a, q)) += t1a[0] * J1[0][q] + t1a[1] * J1[1][q] + t1a[2] * J1[2][q];
              PTS(PT_H + tri(6 + a, 6 + q)) += t2a[0] * J2[0][q] + t2a[1] * J2[1][q] + t2a[2] * J2[2][q];
            }
            PTS(PT_H + tri(6 + a, q)) += t2a[0] * J1[0][q] + t2a[1] * J1[1][q] + t2a[2] * J1[2][q];
          }
        }
      }
    }
    PUSH_TOC(2);
    {   // gradient at round-off / tolerance level: accept the iterate (MuJoCo's scaled-gradient stop)
      double gm = gmax_arm;
      for (int k = 0; k < ncd; k++) gm = fmax(gm, fabs(PTS(PT_P + k)));
      if (gm <= PUSH_GRAD_TOL) { converged = true; break; }
    }
    // ================= phase C: factorise the 12 x 12 cube system, directions
    if (two) {
      double H[78], dc[12], idc[12], pcv[12];
#pragma unroll
      for (int i = 0; i < 78; i++) H[i] = PTS(PT_H + i);
#pragma unroll
      for (int k = 0; k < 12; k++) pcv[k] = PTS(PT_P + k);
      if (!ldl_n<12>(H, dc, idc)) return false;
      ldl_solve_n<12>(H, idc, pcv);
#pragma unroll
      for (int k = 0; k < 12; k++) PTS(PT_P + k) = pcv[k];
    } else {
      double H[21], dc[6], idc[6], pcv[6];
#pragma unroll
      for (int i = 0; i < 21; i++) H[i] = PTS(PT_H + i);
#pragma unroll
      for (int k = 0; k < 6; k++) pcv[k] = PTS(PT_P + k);
      if (!ldl_n<6>(H, dc, idc)) return false;
      ldl_solve_n<6>(H, idc, pcv);
#pragma unroll
      for (int k = 0; k < 6; k++) PTS(PT_P + k) = pcv[k];
    }
    {
      double pa[NDOF];
#pragma unroll
      for (int k = 0; k < NDOF; k++) pa[k] = -PTS(PT_ZG + k);
      if (any_rod) {
        if (rod) {
          double jp3[3], hj[3];
#pragma unroll
          for (int r3 = 0; r3 < 3; r3++) { double a = 0;
#pragma unroll
            for (int k = 0; k < 6; k++) a += PTS(PT_XR + 24 + 6 * r3 + k) * PTS(PT_P + 6 * rod_cube + k); jp3[r3] = a; }
#pragma unroll
          for (int a = 0; a < 3; a++) hj[a] = PTS(PT_XR + 15 + 3 * a) * jp3[0] + PTS(PT_XR + 16 + 3 * a) * jp3[1] + PTS(PT_XR + 17 + 3 * a) * jp3[2];
#pragma unroll
          for (int k = 0; k < NDOF; k++) pa[k] -= PTS(PT_Z + k) * hj[0] + PTS(PT_Z + 9 + k) * hj[1] + PTS(PT_Z + 18 + k) * hj[2];
          // directional derivative of the rod rows
          int base = PT_ROD;
#pragma unroll
          for (int r3 = 0; r3 < 3; r3++) {
            double a = jp3[r3];
#pragma unroll
            for (int k = 0; k < NARM; k++) a += PTS(PT_JA + 7 * r3 + k) * pa[k];
            PTS(base + 14 + r3) = a;
          }
        }
      }
#pragma unroll
      for (int k = 0; k < NDOF; k++) PTS(PT_P + PUSH_ARM0 + k) = pa[k];
    }
    PUSH_TOC(3);
    // ================= phase D: line search quantities
    double pMp = 0, pMa = 0, gTp = 0;
    for (int k = 0; k < PUSH_NV; k++) if (k < ncd || k >= PUSH_ARM0) gTp += PTS(PT_G + k) * PTS(PT_P + k);
    {
      const double Mc[6] = {pc.box_mass, pc.box_mass, pc.box_mass, pc.box_inertia, pc.box_inertia, pc.box_inertia};
      for (int k = 0; k < ncd; k++) { double pk = PTS(PT_P + k); pMp += Mc[k % 6] * pk * pk; pMa += Mc[k % 6] * pk * (PTS(PT_X + k) - PTS(PT_A0 + k)); }
      double Mm[45], pa[NDOF], dx[NDOF], t1[NDOF], t2[NDOF];
#pragma unroll
      for (int i = 0; i < 45; i++) Mm[i] = PTS(PT_M + i);
#pragma unroll
      for (int k = 0; k < NDOF; k++) { pa[k] = PTS(PT_P + PUSH_ARM0 + k); dx[k] = PTS(PT_X + PUSH_ARM0 + k) - PTS(PT_A0 + PUSH_ARM0 + k); }
      symv9(Mm, pa, t1); symv9(Mm, dx, t2);
#pragma unroll
      for (int k = 0; k < NDOF; k++) { pMp += pa[k] * t1[k]; pMa += pa[k] * t2[k]; }
    }
#pragma clang loop unroll(disable)
    for (int b = 0; b < nb; b++) {
      double R[9], pos[3], pb[6];
#pragma unroll
      for (int k = 0; k < 9; k++) R[k] = PTS(PT_R + 9 * b + k);
#pragma unroll
      for (int k = 0; k < 3; k++) pos[k] = PTS(PT_POS + 3 * b + k);
#pragma unroll
      for (int k = 0; k < 6; k++) pb[k] = PTS(PT_P + 6 * b + k);
      CubeFace face;
      cube_face(pc, R, face);
#pragma clang loop unroll(disable)
      for (int i = 0; i < 8; i++) {
        int base = PT_SLAB + 10 * (8 * b + i);
        if (wave_any(PTS(base + 3) != 0)) {
          double r[3], dist;
          slot_geom(pc, face, pos, i, r, &dist);
          double J[3][6]; slab_rows(R, r, J);
#pragma unroll
          for (int rr = 0; rr < 3; rr++) { double a = 0;
#pragma unroll
            for (int k = 0; k < 6; k++) a += J[rr][k] * pb[k]; PTS(base + 7 + rr) = a; }
        }
      }
    }
    for (int ci = 0; ci < nbb_max; ci++) {
      if (ci < nbb) {
        int base = PT_CON + 17 * ci;
        double p[3] = {PTS(base), PTS(base + 1), PTS(base + 2)}, n[3] = {PTS(base + 3), PTS(base + 4), PTS(base + 5)}, t1[3], t2[3];
        make_frame(n, t1, t2);
        double jp3[3] = {0, 0, 0};
        for (int b = 0; b < PUSH_NB; b++) {
          double R[9], r[3], J[3][6];
#pragma unroll
          for (int k = 0; k < 9; k++) R[k] = PTS(PT_R + 9 * b + k);
#pragma unroll
          for (int k = 0; k < 3; k++) r[k] = p[k] - PTS(PT_POS + 3 * b + k);
          box_row_r(R, r, n, J[0]); box_row_r(R, r, t1, J[1]); box_row_r(R, r, t2, J[2]);
          double sg = b == 0 ? -1.0 : 1.0;
#pragma unroll
          for (int r3 = 0; r3 < 3; r3++)
#pragma unroll
            for (int k = 0; k < 6; k++) jp3[r3] += sg * J[r3][k] * PTS(PT_P + 6 * b + k);
        }
        PTS(base + 14) = jp3[0]; PTS(base + 15) = jp3[1]; PTS(base + 16) = jp3[2];
      }
    }
    PUSH_TOC(4);
    // ================= phase E: exact line search
    double alpha = 1, lo = 0, hi = -1, best = 1, wprev = 1e300;
    for (int ls = 0; ls < 40; ls++) {
      D3IL_STAT(g_stats.ls_iters++);
      double d1 = pMa + alpha * pMp, d2 = pMp;
#pragma unroll
      for (int k = NARM; k < NDOF; k++) {   // finger limit rows (arm joints are inside their limits on this path)
        double sign = PTS(PT_LIM + 3 * k), D = PTS(PT_LIM + 3 * k + 1), aref = PTS(PT_LIM + 3 * k + 2);
        double jp = sign * PTS(PT_P + PUSH_ARM0 + k), jar = sign * PTS(PT_X + PUSH_ARM0 + k) - aref + alpha * jp;
        if (sign != 0 && jar < 0) { d1 += D * jar * jp; d2 += D * jp * jp; }
      }
#pragma clang loop unroll(disable)
      for (int i = 0; i < 8 * nb; i++) {
        int base = PT_SLAB + 10 * i;
        double Dn = PTS(base + 3);
        if (wave_any(Dn != 0)) {
          double jp[3] = {PTS(base + 7), PTS(base + 8), PTS(base + 9)};
          double jt[3] = {PTS(base + 4) + alpha * jp[0], PTS(base + 5) + alpha * jp[1], PTS(base + 6) + alpha * jp[2]}, ft[3], Hc[9];
          cone_eval(jt, Dn, Dn * impr, mu0, fric0, ft, Hc);
#pragma unroll
          for (int r = 0; r < 3; r++) { d1 -= ft[r] * jp[r];
#pragma unroll
            for (int q = 0; q < 3; q++) d2 += jp[r] * Hc[3 * r + q] * jp[q]; }
        }
      }
      for (int ci = 0; ci < 9; ci++) {
        if (ci == 8 ? !any_rod : ci >= nbb_max) continue;
        if (ci == 8 ? rod : ci < nbb) {
          int base = PT_CON + 17 * ci;
          double Dn = PTS(base + 10);
          double jp[3] = {PTS(base + 14), PTS(base + 15), PTS(base + 16)};
          double jt[3] = {PTS(base + 11) + alpha * jp[0], PTS(base + 12) + alpha * jp[1], PTS(base + 13) + alpha * jp[2]}, ft[3], Hc[9];
          cone_eval(jt, Dn, Dn * impr, mu1, fric1, ft, Hc);
#pragma unroll
          for (int r = 0; r < 3; r++) { d1 -= ft[r] * jp[r];
#pragma unroll
            for (int q = 0; q < 3; q++) d2 += jp[r] * Hc[3 * r + q] * jp[q]; }
        }
      }
      best = alpha;
      // full Newton step: accepted on the curvature condition phi'(1) <= 0.1 |phi'(0)| (still descending, or just past the minimum)
      if (ls == 0 && d1 <= D3IL_TOL.ls_full * fabs(gTp)) break;
      if (fabs(d1) <= D3IL_TOL.ls_c2 * fabs(gTp) || fabs(d1) <= D3IL_TOL.ls_rel * d2 * alpha || fabs(d1) < 1e-14 * fmax(1.0, fabs(pMa))) break;   // minimiser of phi within 0.1 % of alpha (any such step keeps Newton's rate), or slope at round-off level
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
    PUSH_TOC(5);
    double smax = 0, xmax = 0;
    for (int k = 0; k < PUSH_NV; k++) {
      if (k >= (lane_two ? 12 : 6) && k < PUSH_ARM0) continue;
      double dxk = best * PTS(PT_P + k), xn = PTS(PT_X + k) + dxk;
      PTS(PT_X + k) = xn; smax = fmax(smax, fabs(dxk)); xmax = fmax(xmax, fabs(xn));
    }
    // Newton converges quadratically once the full step is accepted: a step below 1e-6 (relative) leaves an error of the
    // order of its square, so the confirming iteration is skipped
    if (smax <= 1e-12 * (1 + xmax) || (best == 1.0 && smax <= D3IL_TOL.step_rel * (1 + xmax))) converged = true;
  }
  return converged;
}

// general-path collision: slab contacts from the specialised routine, cube <-> cube and rod <-> cube from the general
// ones; writes the contact records; returns the contact count
D3IL_NOINLINE inline int push_collect_contacts(const PushConsts& pc_, const PushScratch sc, const CubeSlab* cs, const double* rodc, const double* rodu,
                                               double rod_r, double rod_h, bool near_bb, const bool* near_rod, unsigned* flags_out, int* has) {
  D3IL_PUSH_CONSTS(pc_, pc);
  int ncon = 0;
  unsigned fl = 0;
  has[0] = has[1] = has[2] = 0;
  for (int b = 0; b < PUSH_NB; b++) for (int i = 0; i < 8; i++) {
    if (!(cs[b].dist[i] < 0)) continue;
    if (ncon >= PUSH_MAXCON) { fl |= PF_CON_OVERFLOW; break; }
    int base = PG_CON + ncon * PREC;
    for (int k = 0; k < 3; k++) { PGS(base + k) = cs[b].r[i][k] + PGS(PG_AUX_POS + 3 * b + k); PGS(base + 3 + k) = k == 2 ? 1.0 : 0.0; }
    PGS(base + 12) = cs[b].dist[i]; PGS(base + 13) = CK_SLAB; PGS(base + 14) = b;
    ncon++;
  }
  double p0[3], p1[3], R0[9], R1[9];
  for (int k = 0; k < 3; k++) { p0[k] = PGS(PG_AUX_POS + k); p1[k] = PGS(PG_AUX_POS + 3 + k); }
  for (int k = 0; k < 9; k++) { R0[k] = PGS(PG_AUX_R + k); R1[k] = PGS(PG_AUX_R + 9 + k); }
  if (near_bb) {
    double rec[8][7];
    int n = box_box(p0, R0, pc.box_half, p1, R1, pc.box_half, 0.0, rec, 8);
    for (int i = 0; i < n; i++) {
      if (ncon >= PUSH_MAXCON) { fl |= PF_CON_OVERFLOW; break; }
      int base = PG_CON + ncon * PREC;
      for (int k = 0; k < 3; k++) { PGS(base + k) = rec[i][1 + k]; PGS(base + 3 + k) = rec[i][4 + k]; }
      PGS(base + 12) = rec[i][0]; PGS(base + 13) = CK_BOXBOX; PGS(base + 14) = 0;
      ncon++; has[0] += 1;
    }
  }
  for (int b = 0; b < PUSH_NB; b++) {
    if (!near_rod[b]) continue;
    double r1[7];
    if (cyl_box(rodc, rodu, rod_r, rod_h, b ? p1 : p0, b ? R1 : R0, pc.box_half, 0.0, r1)) {
      if (ncon >= PUSH_MAXCON) { fl |= PF_CON_OVERFLOW; continue; }
      int base = PG_CON + ncon * PREC;
      for (int k = 0; k < 3; k++) { PGS(base + k) = r1[1 + k]; PGS(base + 3 + k) = r1[4 + k]; }
      PGS(base + 12) = r1[0]; PGS(base + 13) = CK_ROD; PGS(base + 14) = b;
      ncon++; has[1 + b] = 1;
    }
  }
  *flags_out = fl;
  return ncon;
}

// ------------------------------------------------------------------------------------------------ the physics sub-step
D3IL_HD void cube_integrate(BoxState& bx, const double* acc, double h) {
#pragma unroll
  for (int k = 0; k < 6; k++) bx.vel[k] += h * acc[k];
#pragma unroll
  for (int k = 0; k < 3; k++) bx.pos[k] += h * bx.vel[k];
  double w[3] = {bx.vel[3], bx.vel[4], bx.vel[5]}, ang = sqrt(dot3(w, w)) * h;
  if (ang >= 1e-15) {   // mju_quatIntegrate
    double sa = sin(0.5 * ang), ca = cos(0.5 * ang), sc_ = h / ang;
    double dq[4] = {ca, w[0] * sc_ * sa, w[1] * sc_ * sa, w[2] * sc_ * sa}, q[4] = {bx.quat[0], bx.quat[1], bx.quat[2], bx.quat[3]}, r[4];
    r[0] = q[0] * dq[0] - q[1] * dq[1] - q[2] * dq[2] - q[3] * dq[3];
    r[1] = q[0] * dq[1] + q[1] * dq[0] + q[2] * dq[3] - q[3] * dq[2];
    r[2] = q[0] * dq[2] - q[1] * dq[3] + q[2] * dq[0] + q[3] * dq[1];
    r[3] = q[0] * dq[3] + q[1] * dq[2] - q[2] * dq[1] + q[3] * dq[0];
    double nn = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);
#pragma unroll
    for (int k = 0; k < 4; k++) bx.quat[k] = r[k] / nn;
  }
}

// Arm half of one physics sub-step: forward pass, contact-candidate search, the coupled / memory-resident solve when
// something couples, the decoupled arm solve otherwise, arm integration.  box[2]: current cube states; cwarm[12]: the cubes'
// warm start.  Returns a mask: bit b set = cube b took part in a joint solve and its acceleration is in the table at PT_P[6 b ..].
template <class C>
D3IL_HD int push_substep_arm(const C& c0, const PushConsts& pc_, EnvState& st, const BoxState* box, const double* cwarm, const PushScratch sc,
                              const double* tau, const double* ffing) {
  D3IL_PUSH_CONSTS(pc_, pc);
  D3IL_REFRESH(c0, c);
  const double h = c.timestep;
  PUSH_TIC;
  // ---- arm forward pass (panda_step.h physics_substep): dynamics, smooth force, read-backs, factorisation of M
  DynOut dyn;
  dynamics(c0, st.q, st.v, dyn);
  double fs[NDOF];
#pragma unroll
  for (int k = 0; k < NARM; k++) fs[k] = clampd(tau[k] + st.bias[k], c.force_lo[k], c.force_hi[k]) - dyn.bias[k];
#pragma unroll
  for (int k = 0; k < NFING; k++) fs[NARM + k] = clampd(ffing[k], c.force_lo[NARM + k], c.force_hi[NARM + k]) - dyn.bias[NARM + k] - c.f_damping[k] * st.v[NARM + k];
#pragma unroll
  for (int k = 0; k < NARM; k++) st.bias[k] = dyn.bias[k];
  double rodc[3], rodu[3];
  {
    double t[3]; mulE(dyn.R7, c.tcp7, t);
    st.tcp[0] = dyn.p7[0] + t[0]; st.tcp[1] = dyn.p7[1] + t[1]; st.tcp[2] = dyn.p7[2] + t[2];
    mulE(dyn.R7, c.rod_c7, rodc); rodc[0] += dyn.p7[0]; rodc[1] += dyn.p7[1]; rodc[2] += dyn.p7[2];
    mulE(dyn.R7, c.rod_u7, rodu);
  }
  double L[45], d[NDOF], id[NDOF];
  if (!ldl9(dyn.M, L, d, id)) st.flags |= F_SOLVER_FAIL;
  bool arm_rows = false;
#pragma unroll
  for (int k = 0; k < NARM; k++) arm_rows = arm_rows || (st.q[k] - c.jnt_range[k][0] < c.lim_margin[k]) || (c.jnt_range[k][1] - st.q[k] < c.lim_margin[k]);
  // ---- cheap proximity tests decide whether anything can couple the cubes with each other or with the arm
  const double rcirc = sqrt(pc.box_half[0] * pc.box_half[0] + pc.box_half[1] * pc.box_half[1] + pc.box_half[2] * pc.box_half[2]);
  bool near_rod[PUSH_NB], near_bb;
#pragma unroll
  for (int b = 0; b < PUSH_NB; b++) {
    if (fabs(box[b].pos[0] - pc.slab_c[0][0]) > pc.slab_h[0][0] - 0.06 || fabs(box[b].pos[1] - pc.slab_c[0][1]) > pc.slab_h[0][1] - 0.06) st.flags |= PF_OFF_TABLE;
    double w[3] = {box[b].pos[0] - rodc[0], box[b].pos[1] - rodc[1], box[b].pos[2] - rodc[2]};
    double t = clampd(dot3(w, rodu), -c.rod_h, c.rod_h);
    double e[3] = {w[0] - t * rodu[0], w[1] - t * rodu[1], w[2] - t * rodu[2]};
    near_rod[b] = dot3(e, e) < (rcirc + c.rod_r) * (rcirc + c.rod_r);
  }
  {
    double dd[3] = {box[1].pos[0] - box[0].pos[0], box[1].pos[1] - box[0].pos[1], box[1].pos[2] - box[0].pos[2]};
    near_bb = dot3(dd, dd) < 4 * rcirc * rcirc;
    if (wave_any(near_bb)) {
      // inside the circumscribed spheres: the six face axes of the separating-axis test (the first part of box_box) decide
      // most cases without the general routine; a positive separation on any of them means no contact (margin 0)
      double R0[9], R1[9];
      quat2mat(box[0].quat, R0); quat2mat(box[1].quat, R1);
      bool sep = false;
#pragma unroll
      for (int i = 0; i < 3; i++) {
        double a0[3] = {R0[i], R0[3 + i], R0[6 + i]}, a1[3] = {R1[i], R1[3 + i], R1[6 + i]};
        double e0 = 0, e1 = 0;
#pragma unroll
        for (int j = 0; j < 3; j++) {
          double b1[3] = {R1[j], R1[3 + j], R1[6 + j]}, b0[3] = {R0[j], R0[3 + j], R0[6 + j]};
          e0 += pc.box_half[j] * fabs(dot3(a0, b1)); e1 += pc.box_half[j] * fabs(dot3(a1, b0));
        }
        sep = sep || fabs(dot3(dd, a0)) - (pc.box_half[i] + e0) > 0 || fabs(dot3(dd, a1)) - (pc.box_half[i] + e1) > 0;
      }
      near_bb = near_bb && !sep;
    }
  }
  double fc[NDOF];
#pragma unroll
  for (int k = 0; k < NDOF; k++) fc[k] = 0;
  const bool general = arm_rows || near_bb || near_rod[0] || near_rod[1];
  bool solved = false;
  int solved_mask = 0;      // bit b: cube b's acceleration comes from the joint solve (table, PT_P)
  PUSH_TOC(6);
  if (wave_any(general)) {
    if (general) {
      // ---- general path: publish the inputs to the scratch area, collect contacts, solve if anything couples
      unsigned cfl = 0; int has[3], ncon;
      {
        double Rb[PUSH_NB][9];
        CubeSlab cs[PUSH_NB];
#pragma unroll
        for (int b = 0; b < PUSH_NB; b++) {
          quat2mat(box[b].quat, Rb[b]);
          cube_slab_contacts(pc, box[b].pos, Rb[b], cs[b]);
#pragma unroll
          for (int k = 0; k < 9; k++) PGS(PG_AUX_R + 9 * b + k) = Rb[b][k];
#pragma unroll
          for (int k = 0; k < 3; k++) PGS(PG_AUX_POS + 3 * b + k) = box[b].pos[k];
#pragma unroll
          for (int k = 0; k < 6; k++) PGS(PG_AUX_VEL + 6 * b + k) = box[b].vel[k];
        }
        ncon = push_collect_contacts(pc, sc, cs, rodc, rodu, c.rod_r, c.rod_h, near_bb, near_rod, &cfl, has);
      }
      st.flags |= cfl;
      if (has[0] || has[1] || has[2] || arm_rows) {
        solved = true;   // the arm takes part in a joint solve
        const bool slow = arm_rows || (has[1] && has[2]);   // arm joint at a limit, or the rod on both cubes: memory-resident solver
        double a0[NDOF];
#pragma unroll
        for (int k = 0; k < NDOF; k++) a0[k] = fs[k];
        ldl9_solve(L, id, a0);
        if (wave_any(!slow)) {
          if (!slow) {
            // ---- coupled path: fill the LDS table.  Cube groups: group g holds cube g ^ perm; without a cube-cube contact the
            // cube under the rod becomes group 0 so that a wave without any cube-cube contact processes one group only
            const int perm = (has[0] == 0 && has[2]) ? 1 : 0;
            const bool two = wave_any(has[0] > 0);
#pragma unroll
            for (int g = 0; g < PUSH_NB; g++) {
              const BoxState& bg = (g ^ perm) ? box[1] : box[0];
#pragma unroll
              for (int k = 0; k < 9; k++) PTS(PT_R + 9 * g + k) = PGS(PG_AUX_R + 9 * (g ^ perm) + k);
#pragma unroll
              for (int k = 0; k < 3; k++) PTS(PT_POS + 3 * g + k) = bg.pos[k];
#pragma unroll
              for (int k = 0; k < 6; k++) { PTS(PT_VEL + 6 * g + k) = bg.vel[k]; PTS(PT_A0 + 6 * g + k) = k < 3 ? c.gravity[k] : 0.0; }
            }
#pragma unroll
            for (int k = 0; k < NDOF; k++) { PTS(PT_VEL + PUSH_ARM0 + k) = st.v[k]; PTS(PT_A0 + PUSH_ARM0 + k) = a0[k]; }
#pragma unroll
            for (int i = 0; i < 45; i++) PTS(PT_M + i) = dyn.M[i];
#pragma unroll
            for (int k = 0; k < NDOF; k++) {
              double sign = 0, D = 0, aref = 0;
              if (k >= NARM) {   // arm joints are inside their limits on this path
                double dlo = st.q[k] - c.jnt_range[k][0], dhi = c.jnt_range[k][1] - st.q[k], dist = 0;
                if (dlo < c.lim_margin[k]) { sign = 1; dist = dlo; }
                else if (dhi < c.lim_margin[k]) { sign = -1; dist = dhi; }
                if (sign != 0) {
                  double imp = impedance(c.lim_solimp[k], dist - c.lim_margin[k]);
                  D = 1 / fmax(1e-15, (1 - imp) / imp * c.dof_invweight0[k]);
                  aref = -c.lim_B[k] * (sign * st.v[k]) - c.lim_K[k] * imp * (dist - c.lim_margin[k]);
                }
              }
              PTS(PT_LIM + 3 * k) = sign; PTS(PT_LIM + 3 * k + 1) = D; PTS(PT_LIM + 3 * k + 2) = aref;
            }
            int nbb = 0, rod_cube = -1;
            for (int ci = 0; ci < ncon; ci++) {
              int base = PG_CON + ci * PREC, kind = (int)PGS(base + 13);
              if (kind == CK_SLAB) continue;
              int dst = kind == CK_BOXBOX ? PT_CON + 17 * nbb : PT_ROD;
              for (int k = 0; k < 6; k++) PTS(dst + k) = PGS(base + k);
              PTS(dst + 6) = PGS(base + 12);
              if (kind == CK_BOXBOX) nbb++;
              else {
                rod_cube = (int)PGS(base + 14) ^ perm;
                double R7[9], p7[3], ax[NARM][3], og[NARM][3];
                world_chain(c0, dyn.sn, dyn.cs, R7, p7, ax, og);
                double p[3] = {PGS(base), PGS(base + 1), PGS(base + 2)}, n[3] = {PGS(base + 3), PGS(base + 4), PGS(base + 5)}, t1[3], t2[3];
                make_frame(n, t1, t2);
#pragma unroll
                for (int k = 0; k < NARM; k++) {
                  double dd[3] = {p[0] - og[k][0], p[1] - og[k][1], p[2] - og[k][2]}, col[3];
                  cross3(ax[k], dd, col);
                  PTS(PT_JA + k) = dot3(n, col); PTS(PT_JA + 7 + k) = dot3(t1, col); PTS(PT_JA + 14 + k) = dot3(t2, col);
                }
              }
            }
            if (st.flags & PF_WARM_VALID) for (int k = 0; k < PUSH_NV; k++) PTS(PT_X + k) = k < PUSH_ARM0 ? cwarm[6 * ((k / 6) ^ perm) + k % 6] : PWS(k);
            else for (int k = 0; k < PUSH_NV; k++) PTS(PT_X + k) = PTS(PT_A0 + k);
            if (has[0] == 0) for (int k = 6; k < 12; k++) PTS(PT_X + k) = PTS(PT_A0 + k);   // group 1 is not part of this lane's system
            if (!coupled_newton(pc, sc, nbb, rod_cube, c.rod_invweight0, two)) st.flags |= F_SOLVER_FAIL;
            for (int k = PUSH_ARM0; k < PUSH_NV; k++) PWS(k) = PTS(PT_X + k);
            // hand the cube accelerations over in physical cube order (PT_P is free again)
            for (int b = 0; b < PUSH_NB; b++) {
              const int g = b ^ perm;
              if (g == 0 || has[0] > 0) { solved_mask |= 1 << b; for (int k = 0; k < 6; k++) PTS(PT_P + 6 * b + k) = PTS(PT_X + 6 * g + k); }
            }
          }
        }
        if (slow) {
#pragma unroll
          for (int k = 0; k < NDOF; k++) PGS(PG_AUX_VEL + PUSH_ARM0 + k) = st.v[k];
#pragma unroll
          for (int i = 0; i < 45; i++) PGS(PG_M + i) = dyn.M[i];
#pragma unroll
          for (int k = 0; k < NDOF; k++) PGS(PG_A0 + PUSH_ARM0 + k) = a0[k];
#pragma unroll
          for (int b = 0; b < PUSH_NB; b++)
#pragma unroll
            for (int k = 0; k < 6; k++) PGS(PG_A0 + 6 * b + k) = k < 3 ? c.gravity[k] : 0.0;
#pragma unroll
          for (int k = 0; k < NDOF; k++) {   // limit rows of all nine joints
            double dlo = st.q[k] - c.jnt_range[k][0], dhi = c.jnt_range[k][1] - st.q[k];
            double sign = 0, dist = 0, D = 0, aref = 0;
            if (dlo < c.lim_margin[k]) { sign = 1; dist = dlo; }
            else if (dhi < c.lim_margin[k]) { sign = -1; dist = dhi; }
            if (sign != 0) {
              double imp = impedance(c.lim_solimp[k], dist - c.lim_margin[k]);
              D = 1 / fmax(1e-15, (1 - imp) / imp * c.dof_invweight0[k]);
              aref = -c.lim_B[k] * (sign * st.v[k]) - c.lim_K[k] * imp * (dist - c.lim_margin[k]);
            }
            PGS(PG_AUX_LIM + 3 * k) = sign; PGS(PG_AUX_LIM + 3 * k + 1) = D; PGS(PG_AUX_LIM + 3 * k + 2) = aref;
          }
          PGS(PG_AUX_LIM + 27) = c.rod_invweight0;
          if (has[1] || has[2]) {   // arm Jacobian rows of the rod contacts
            double R7[9], p7[3], ax[NARM][3], og[NARM][3];
            world_chain(c0, dyn.sn, dyn.cs, R7, p7, ax, og);
            for (int ci = 0; ci < ncon; ci++) {
              int base = PG_CON + ci * PREC;
              if ((int)PGS(base + 13) != CK_ROD) continue;
              int b = (int)PGS(base + 14);
              double p[3] = {PGS(base), PGS(base + 1), PGS(base + 2)}, n[3] = {PGS(base + 3), PGS(base + 4), PGS(base + 5)}, t1[3], t2[3];
              make_frame(n, t1, t2);
#pragma unroll
              for (int k = 0; k < NARM; k++) {
                double dd[3] = {p[0] - og[k][0], p[1] - og[k][1], p[2] - og[k][2]}, col[3];
                cross3(ax[k], dd, col);
                PGS(PG_JA + b * 21 + k) = dot3(n, col); PGS(PG_JA + b * 21 + 7 + k) = dot3(t1, col); PGS(PG_JA + b * 21 + 14 + k) = dot3(t2, col);
              }
            }
          }
          if (st.flags & PF_WARM_VALID) for (int k = 0; k < PUSH_NV; k++) PGS(PG_X + k) = k < PUSH_ARM0 ? cwarm[k] : PWS(k);
          else for (int k = 0; k < PUSH_NV; k++) PGS(PG_X + k) = PGS(PG_A0 + k);
          if (!push_general_solve(pc, sc, ncon, has[0] != 0, has[1] != 0, has[2] != 0)) st.flags |= F_SOLVER_FAIL;
          for (int k = 0; k < PUSH_NV; k++) { double xk = PGS(PG_X + k); if (k < PUSH_ARM0) PTS(PT_P + k) = xk; else { PTS(PT_X + k) = xk; PWS(k) = xk; } }
          solved_mask = 3;
        }
        {   // constraint force on the arm from the optimality condition M (x - a0) = J' f
          double xa[NDOF], Mx[NDOF];
#pragma unroll
          for (int k = 0; k < NDOF; k++) xa[k] = PWS(PUSH_ARM0 + k);
          symv9(dyn.M, xa, Mx);
#pragma unroll
          for (int k = 0; k < NDOF; k++) fc[k] = Mx[k] - fs[k];
        }
      }
    }
  }
  PUSH_TOC(7);
  if (!solved) {
    // ---- decoupled path, arm: finger-limit rows by the exact active-set solution (panda_step.h)
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
    if (fsign[0] != 0 || fsign[1] != 0) {
      double l87 = L[tri(8, 7)];
      double W00 = id[7] + l87 * l87 * id[8], W01 = -l87 * id[8], W11 = id[8];
      double a0[NDOF];
#pragma unroll
      for (int k = 0; k < NDOF; k++) a0[k] = fs[k];
      ldl9_solve(L, id, a0);
      double s0 = fsign[0], s1 = fsign[1];
      double r0 = s0 * a0[7] - faref[0], r1 = s1 * a0[8] - faref[1];
      double G00 = W00 * s0 * s0, G01 = W01 * s0 * s1, G11 = W11 * s1 * s1;
      double f0 = 0, f1 = 0;
      bool have0 = s0 != 0, have1 = s1 != 0, done = false;
      if (have0 && have1) {
        double a = 1 + fD[0] * G00, b = fD[0] * G01, cc = fD[1] * G01, dd = 1 + fD[1] * G11;
        double det = a * dd - b * cc, y0 = -fD[0] * r0, y1 = -fD[1] * r1;
        double g0 = (dd * y0 - b * y1) / det, g1 = (a * y1 - cc * y0) / det;
        if (g0 > 0 && g1 > 0) { f0 = g0; f1 = g1; done = true; }
      }
      if (!done && have0) {
        double g0 = -fD[0] * r0 / (1 + fD[0] * G00);
        if (g0 > 0 && (!have1 || r1 + G01 * g0 >= 0)) { f0 = g0; f1 = 0; done = true; }
      }
      if (!done && have1) {
        double g1 = -fD[1] * r1 / (1 + fD[1] * G11);
        if (g1 > 0 && (!have0 || r0 + G01 * g1 >= 0)) { f1 = g1; f0 = 0; done = true; }
      }
      fc[7] = s0 * f0; fc[8] = s1 * f1;
    }
  }
  // ---- arm: semi-implicit Euler, (M + h B) qacc = qfrc_smooth + qfrc_constraint with B on the fingers (last two pivots)
  {
    double hb0 = h * c.f_damping[0], hb1 = h * c.f_damping[1];
    double l87 = L[tri(8, 7)];
    double S11 = d[8] + l87 * l87 * d[7];
    double d7n = d[7] + hb0, i7 = rcpd(d7n);
    double l87n = l87 * d[7] * i7;
    double d8n = S11 + hb1 - l87n * l87n * d7n;
    d[7] = d7n; d[8] = d8n; id[7] = i7; id[8] = rcpd(d8n); L[tri(8, 7)] = l87n;
    double qacc[NDOF];
#pragma unroll
    for (int k = 0; k < NDOF; k++) qacc[k] = fs[k] + fc[k];
    ldl9_solve(L, id, qacc);
#pragma unroll
    for (int k = 0; k < NDOF; k++) { st.v[k] += h * qacc[k]; st.q[k] += h * st.v[k]; }
    if (!solved)
#pragma unroll
      for (int k = 0; k < NDOF; k++) PWS(PUSH_ARM0 + k) = qacc[k];
  }
  PUSH_TOC(8);
  return solved_mask;
}

// Cube half: one free cube integrates either with its share of the joint solution (table, PT_X) or with its own
// decoupled 6-dof solve over the slab contacts.  warm[6]: this cube's warm start in / solution out.
D3IL_HD void push_substep_cube(const PushConsts& pc_, const double* gravity, double h, BoxState& bx, double* warm, int b, bool solved, bool warm_valid,
                               unsigned& flags, const PushScratch sc) {
  D3IL_PUSH_CONSTS(pc_, pc);
  PUSH_TIC;
  double xb[6];
  if (!solved) {
    double Rb[9], a0b[6] = {gravity[0], gravity[1], gravity[2], 0, 0, 0}, vb[6];
    CubeSlab cs;
    quat2mat(bx.quat, Rb);
    cube_slab_contacts(pc, bx.pos, Rb, cs);
#pragma unroll
    for (int k = 0; k < 6; k++) { xb[k] = warm_valid ? warm[k] : a0b[k]; vb[k] = bx.vel[k]; }
    if (!cube_newton(pc, Rb, vb, cs, a0b, xb)) flags |= F_SOLVER_FAIL;
  } else {
#pragma unroll
    for (int k = 0; k < 6; k++) xb[k] = PTS(PT_P + 6 * b + k);
  }
#pragma unroll
  for (int k = 0; k < 6; k++) warm[k] = xb[k];
  cube_integrate(bx, xb, h);
  PUSH_TOC(9);
}

// One physics sub-step on a single lane (host build, reset kernel): arm half, then both cubes in turn.  The step kernel
// runs the two cube halves on two lanes instead (push_kernels.h).
template <class C>
D3IL_HD void push_physics_substep(const C& c0, const PushConsts& pc_, PushState& ps, const PushScratch sc, const double* tau, const double* ffing) {
  D3IL_PUSH_CONSTS(pc_, pc);
  double cw[12];
  for (int k = 0; k < 12; k++) cw[k] = PWS(k);
  const bool warm_valid = (ps.arm.flags & PF_WARM_VALID) != 0;
  const int solved_mask = push_substep_arm(c0, pc, ps.arm, ps.box, cw, sc, tau, ffing);
  D3IL_REFRESH(c0, c);
  const double grav[3] = {c.gravity[0], c.gravity[1], c.gravity[2]};
#pragma unroll
  for (int b = 0; b < PUSH_NB; b++) {
    double w6[6];
#pragma unroll
    for (int k = 0; k < 6; k++) w6[k] = cw[6 * b + k];
    push_substep_cube(pc, grav, c.timestep, ps.box[b], w6, b, ((solved_mask >> b) & 1) != 0, warm_valid, ps.arm.flags, sc);
#pragma unroll
    for (int k = 0; k < 6; k++) PWS(6 * b + k) = w6[k];
  }
  ps.arm.flags |= PF_WARM_VALID;
}

// ------------------------------------------------------------------------------------------------ task logic (pushing.py)
D3IL_HD double push_tan_yaw(const double* q) {   // np.tan(quat2euler(q)[-1]); geometric_transformation.py:92-111,165-188
  const double FEPS = 2.220446049250313e-16;
  double w = q[0], x = q[1], y = q[2], z = q[3], Nq = w * w + x * x + y * y + z * z;
  double m00 = 1, m01 = 0, m10 = 0, m11 = 1, m12 = 0, m22 = 1;
  if (Nq > FEPS) {
    double s = 2.0 / Nq, X = x * s, Y = y * s, Z = z * s;
    double wX = w * X, wZ = w * Z, xX = x * X, xY = x * Y, yY = y * Y, yZ = y * Z, zZ = z * Z;
    m00 = 1.0 - (yY + zZ); m01 = xY - wZ; m10 = xY + wZ; m11 = 1.0 - (xX + zZ); m12 = yZ - wX; m22 = 1.0 - (xX + yY);
  }
  double cy = sqrt(m22 * m22 + m12 * m12);
  double yaw = cy > 4 * FEPS ? -atan2(m01, m00) : -atan2(-m10, m11);
  return tan(yaw);
}
D3IL_HD void push_dists(const PushConsts& pc_, const PushState& ps, double* d) {   // rr rg gr gg
  D3IL_PUSH_CONSTS(pc_, pc);
  for (int b = 0; b < 2; b++) for (int t = 0; t < 2; t++) {
    double dx = ps.box[b].pos[0] - pc.target[t][0], dy = ps.box[b].pos[1] - pc.target[t][1], dz = ps.box[b].pos[2] - pc.target[t][2];
    d[2 * b + t] = sqrt(dx * dx + dy * dy + dz * dz);
  }
}
D3IL_HD bool push_success(const PushConsts& pc_, const PushState& ps) {   // pushing.py:440-459
  D3IL_PUSH_CONSTS(pc_, pc);
  double d[4]; push_dists(pc, ps, d);
  return (d[0] <= pc.min_dist && d[3] <= pc.min_dist) || (d[1] <= pc.min_dist && d[2] <= pc.min_dist);
}
D3IL_HD void push_obs(const PushState& ps, float* obs) {   // pushing.py:255-280
  obs[0] = (float)ps.arm.tcp[0]; obs[1] = (float)ps.arm.tcp[1];
  for (int b = 0; b < 2; b++) { obs[2 + 3 * b] = (float)ps.box[b].pos[0]; obs[3 + 3 * b] = (float)ps.box[b].pos[1]; obs[4 + 3 * b] = (float)push_tan_yaw(ps.box[b].quat); }
}
// before the physics of a step: obs, reward, done (gym_env_wrapper.py:88-90,124-137)
D3IL_HD void push_step_begin(const PushConsts& pc_, PushState& ps, float* obs, double* reward, unsigned char* done, int max_steps) {
  D3IL_PUSH_CONSTS(pc_, pc);
  push_obs(ps, obs);
  double d[4]; push_dists(pc, ps, d);
  double dx = ps.arm.tcp[0] - ps.box[0].pos[0], dy = ps.arm.tcp[1] - ps.box[0].pos[1];
  *reward = -(sqrt(dx * dx + dy * dy) + d[0]);   // get_reward, pushing.py:379-407
  bool fin = (ps.arm.flags & F_TERMINATED) != 0;
  if (!fin && push_success(pc, ps)) { ps.arm.flags |= F_TERMINATED; fin = true; }
  if (!fin && ps.arm.step >= max_steps - 1) fin = true;
  *done = fin ? 1 : 0;
}
// after the physics: success, first-visit mode logic (pushing.py:335-377)
D3IL_HD void push_step_end(const PushConsts& pc_, PushState& ps, double* mean_distance) {
  D3IL_PUSH_CONSTS(pc_, pc);
  ps.arm.step++;
  double d[4]; push_dists(pc, ps, d);
  double md = pc.min_dist;
  bool succ = (d[0] <= md && d[3] <= md) || (d[1] <= md && d[2] <= md);
  ps.arm.flags &= ~F_SUCCESS;
  if (succ) ps.arm.flags |= F_SUCCESS | F_TERMINATED;
  int first = (int)(ps.arm.flags & PF_FIRST_MASK) - 1, visit = -1, mode = -1;
  if (d[0] <= md && first != 0) visit = 0;
  else if (d[1] <= md && first != 1) visit = 1;
  else if (d[2] <= md && first != 2) visit = 2;
  else if (d[3] <= md && first != 3) visit = 3;
  if (first == -1) first = visit;
  else {
    if (first == 0 && visit == 3) mode = 0;
    else if (first == 3 && visit == 0) mode = 1;
    else if (first == 1 && visit == 2) mode = 2;
    else if (first == 2 && visit == 1) mode = 3;
  }
  ps.arm.flags = (ps.arm.flags & ~(PF_FIRST_MASK | PF_MODE_MASK)) | (unsigned)(first + 1) | ((unsigned)(mode + 1) << PF_MODE_SHIFT);
  *mean_distance = 0.5 * (fmin(d[0], d[1]) + fmin(d[2], d[3]));
}

// ------------------------------------------------------------------------------------------------ env level
// joint PD on the set-point + finger PD (Scene.next_step after the IK update): arm torque without gravity compensation, raw finger force
template <class C>
D3IL_HD void push_control(const C& c, const EnvState& st, const double* q_des, const double* qd_des, double set_width, bool grasp, double* tau, double* ff) {
#pragma unroll
  for (int k = 0; k < NARM; k++) tau[k] = c.pd_p[k] * (q_des[k] - st.q[k]) + c.pd_d[k] * (qd_des[k] - st.v[k]);
  double mean = 0.5 * (st.q[NARM] + st.q[NARM + 1]);   // RobotBase.fing_ctrl_step (Robots.py:441-476)
#pragma unroll
  for (int k = 0; k < NFING; k++) {
    double w = st.q[NARM + k], wv = st.v[NARM + k];
    double f1 = 500 * (mean - w), f2;
    if (mean - set_width > 0.005) f2 = grasp ? -20.0 : 10 * (-0.2 - wv);
    else f2 = clampd(500 * (set_width - w) - 10 * wv, -5, 5);
    ff[k] = f1 + f2;
  }
}
template <class C>
D3IL_HD void push_control_and_physics(const C& c, const PushConsts& pc_, PushState& ps, const PushScratch sc, const double* q_des, const double* qd_des,
                                      double set_width, bool grasp) {
  D3IL_PUSH_CONSTS(pc_, pc);
  double tau[NARM], ff[NFING];
  push_control(c, ps.arm, q_des, qd_des, set_width, grasp, tau, ff);
  push_physics_substep(c, pc, ps, sc, tau, ff);
}

// Block_Push_Env.reset(random=False, context) (pushing.py:461-483): scene.reset, beam to init_qpos, context written into
// the cubes' qpos (z = 0, pushing.py:99-113), one PD-hold sub-step.  ctx = 2 x (pos3, quat4).
template <class C>
D3IL_HD void push_env_reset(const C& c, const PushConsts& pc_, PushState& ps, const PushScratch sc, const double* init_qpos, const double* ctx, float* obs) {
  D3IL_PUSH_CONSTS(pc_, pc);
  EnvState& st = ps.arm;
#pragma unroll
  for (int k = 0; k < NARM; k++) { st.q[k] = init_qpos[k]; st.ikq[k] = 0; st.ikqd[k] = 0; }
  st.q[NARM] = 0; st.q[NARM + 1] = 0;
#pragma unroll
  for (int k = 0; k < NDOF; k++) st.v[k] = 0;
  st.flags = 0; st.step = 0;
  for (int b = 0; b < PUSH_NB; b++) {
    for (int k = 0; k < 3; k++) ps.box[b].pos[k] = ctx[7 * b + k];
    for (int k = 0; k < 4; k++) ps.box[b].quat[k] = ctx[7 * b + 3 + k];
    for (int k = 0; k < 6; k++) ps.box[b].vel[k] = 0;
  }
  for (int k = 0; k < PUSH_NV; k++) PWS(k) = 0;
  {
    DynOut dyn;
    dynamics(c, st.q, st.v, dyn);
#pragma unroll
    for (int k = 0; k < NARM; k++) st.bias[k] = dyn.bias[k];
  }
  double zero[NARM] = {0, 0, 0, 0, 0, 0, 0};
  push_control_and_physics(c, pc, ps, sc, init_qpos, zero, 0.001, false);
  push_obs(ps, obs);
}

// Block_Push_Env.step (pushing.py:335-339) over GymEnvWrapper.step (gym_env_wrapper.py:45-100), one lane doing both the
// controller and the physics (the split-wave kernel in rollout.hip runs the same pieces on two waves)
template <bool FAST, class C>
D3IL_HD void push_env_step(const C& c, const PushConsts& pc_, PushState& ps, const PushScratch sc, const double* action, float* obs, double* reward,
                           unsigned char* done, double* mean_distance, int n_substeps, int max_steps) {
  D3IL_PUSH_CONSTS(pc_, pc);
  push_step_begin(pc, ps, obs, reward, done, max_steps);
  double des[7];
  make_setpoint(action, des);
  double vwarm[7]; vwarm[6] = 0.0;
#pragma clang loop unroll(disable)
  for (int s = 0; s < n_substeps; s++) {
    D3IL_REFRESH(c, cs);
    ik_update<FAST>(cs, des, des + 3, ps.arm.q, ps.arm.flags, ps.arm.ikq, ps.arm.ikqd, vwarm);
    push_control_and_physics(cs, pc, ps, sc, ps.arm.ikq, ps.arm.ikqd, 0.04, false);
  }
  push_step_end(pc, ps, mean_distance);
}

// ------------------------------------------------------------------------------------------------ constants from the blob
#if defined(__HIPCC__)
#define D3IL_HOSTFN __host__
#else
#define D3IL_HOSTFN
#endif
D3IL_HOSTFN inline int build_push_consts(const d3il_model_blob& m, PushConsts& pc, const char** err) {
  std::memset(&pc, 0, sizeof pc);
  if (m.n_obj != PUSH_NB) { *err = "pushing needs two task objects"; return -1; }
  auto geom_of = [&](int body) { for (int g = 0; g < m.ngeom; g++) if (m.geom_body[g] == body && m.geom_contype[g]) return g; return -1; };
  auto body_named_geom = [&](double hx, double hy, double hz) {
    for (int g = 0; g < m.ngeom; g++) if (m.geom_type[g] == D3IL_GEOM_BOX && std::fabs(m.geom_size[g][0] - hx) < 1e-12 && std::fabs(m.geom_size[g][1] - hy) < 1e-12 && std::fabs(m.geom_size[g][2] - hz) < 1e-12) return g;
    return -1;
  };
  int gb[2] = {geom_of(m.obj_body[0]), geom_of(m.obj_body[1])};
  if (gb[0] < 0 || gb[1] < 0 || m.geom_type[gb[0]] != D3IL_GEOM_BOX) { *err = "task objects must be boxes"; return -1; }
  for (int k = 0; k < 3; k++) {
    pc.box_half[k] = m.geom_size[gb[0]][k];
    if (m.geom_size[gb[1]][k] != pc.box_half[k] || m.geom_pos[gb[0]][k] != 0 || m.geom_pos[gb[1]][k] != 0) { *err = "cubes must be identical and centred on their bodies"; return -1; }
  }
  int b0 = m.obj_body[0];
  pc.box_mass = m.body_mass[b0]; pc.box_inertia = m.body_inertia[b0][0];
  if (std::fabs(m.body_inertia[b0][1] - pc.box_inertia) > 1e-15 || std::fabs(m.body_inertia[b0][2] - pc.box_inertia) > 1e-15 ||
      m.body_mass[m.obj_body[1]] != pc.box_mass || m.body_inertia[m.obj_body[1]][0] != pc.box_inertia) { *err = "cube inertia must be isotropic and equal"; return -1; }
  pc.box_invw_t = 1.0 / pc.box_mass; pc.box_invw_r = 1.0 / pc.box_inertia;   // free body: invweight0 = diag(M^-1) averages
  // the two static slabs under the cubes: table_plane (0.49 0.98 0.001) and support_body (0.49 0.98 0.4), lab_surrounding.xml:3-4,112-114
  int gs[2] = {body_named_geom(0.49, 0.98, 0.001), body_named_geom(0.49, 0.98, 0.4)};
  if (gs[0] < 0 || gs[1] < 0) { *err = "table slabs not found"; return -1; }
  for (int s = 0; s < 2; s++) {
    // world pose of a static body chain (all identity orientations in this scene)
    double p[3] = {m.geom_pos[gs[s]][0], m.geom_pos[gs[s]][1], m.geom_pos[gs[s]][2]};
    for (int b = m.geom_body[gs[s]]; b > 0; b = m.body_parent[b]) {
      if (m.body_quat[b][0] != 1.0 || m.body_jntnum[b] != 0) { *err = "slab must be static and axis aligned"; return -1; }
      for (int k = 0; k < 3; k++) p[k] += m.body_pos[b][k];
    }
    for (int k = 0; k < 3; k++) { pc.slab_c[s][k] = p[k]; pc.slab_h[s][k] = m.geom_size[gs[s]][k]; }
  }
  // mixed contact parameters (mj_contactParam, equal priority and solmix): solref/solimp averaged, friction = max
  auto mix = [&](int g1, int g2, int set) {
    double sr[2], si[5];
    for (int k = 0; k < 2; k++) sr[k] = 0.5 * (m.geom_solref[g1][k] + m.geom_solref[g2][k]);
    for (int k = 0; k < 5; k++) si[k] = 0.5 * (m.geom_solimp[g1][k] + m.geom_solimp[g2][k]);
    double dmax = std::fmin(0.9999, std::fmax(0.0001, si[1])), tc = std::fmax(sr[0], 2 * m.timestep);
    pc.ct_K[set] = 1 / std::fmax(1e-15, dmax * dmax * tc * tc * sr[1] * sr[1]);
    pc.ct_B[set] = 2 / std::fmax(1e-15, dmax * tc);
    for (int k = 0; k < 5; k++) pc.ct_solimp[set][k] = si[k];
    pc.ct_solimp[set][0] = std::fmin(0.9999, std::fmax(0.0001, si[0])); pc.ct_solimp[set][1] = dmax;
    pc.ct_fric[set] = std::fmax(m.geom_friction[g1][0], m.geom_friction[g2][0]);
  };
  mix(gs[0], gb[0], 0);
  mix(gb[0], gb[1], 1);
  // the rod and the second slab must mix to the same sets
  if (m.rod_geom < 0) { *err = "no rod geom"; return -1; }
  for (int k = 0; k < 2; k++) if (m.geom_solref[m.rod_geom][k] != m.geom_solref[gb[0]][k] || m.geom_solref[gs[1]][k] != m.geom_solref[gs[0]][k]) { *err = "unexpected contact parameters"; return -1; }
  for (int k = 0; k < 5; k++) if (m.geom_solimp[m.rod_geom][k] != m.geom_solimp[gb[0]][k] || m.geom_solimp[gs[1]][k] != m.geom_solimp[gs[0]][k]) { *err = "unexpected contact parameters"; return -1; }
  for (int k = 0; k < 3; k++) { pc.target[0][k] = m.task_f[k]; pc.target[1][k] = m.task_f[3 + k]; }
  pc.min_dist = m.task_f[6];
  pc.impratio = m.impratio;
  return 0;
}

}  // namespace d3il
