// rigid_common.h - pieces shared by the contact engines (generic engine gen_step.h: Sorting / Inserting / Pushing; wave-cooperative engine stack_step.h:
// Stacking / Aligning): device-side timers of the diagnostics build, the stopping rule of the contact Newton solvers, flag bits, the per-lane scratch views,
// box <-> box and cylinder <-> box collision (MuJoCo's analytic box-box collider and a capsule-style rod test), contact rows, the elliptic
// cone, small LDL^T helpers, free-body integration, the joint PD + finger PD control law (Controller.py:164-185, Robots.py:441-476).
// Everything here compiles for the host as well (tests/hostcheck).  The round-1 Pushing engine that used to live in this file (push_step.h / push_kernels.h)
// is gone: Pushing runs on the generic engine.
#pragma once
#include "panda_step.h"

#if defined(D3IL_DEVICE_STATS) && defined(__HIP_DEVICE_COMPILE__)
// -DD3IL_STATS_PER_WAVE (tools/gpu_sort_phases.py --per-wave): the rows of the timer table are WAVES (4 blockIdx + wave) instead of workgroups, and row 4 blockIdx + 3
// holds, per wave of the generic engine's step kernel, the ticks it waited at the workgroup barrier of a sub-step (gen_kernels.h)
#if defined(D3IL_STATS_PER_WAVE)
#define PUSH_ROW (blockIdx.x * 4 + (threadIdx.x >> 6))
#define PUSH_ROW_OK (blockIdx.x < 1024)
#else
#define PUSH_ROW blockIdx.x
#define PUSH_ROW_OK (blockIdx.x < 4096)
#endif
#define PUSH_TIC unsigned long long t_tic_ = wall_clock64()
#define PUSH_TOC(slot) do { unsigned long long t_now_ = wall_clock64(); \
    if (__builtin_amdgcn_mbcnt_hi(__builtin_amdgcn_read_exec_hi(), __builtin_amdgcn_mbcnt_lo(__builtin_amdgcn_read_exec_lo(), 0u)) == 0 && PUSH_ROW_OK) \
      atomicAdd(&d3il::g_dev_wave[PUSH_ROW][slot], t_now_ - t_tic_); t_tic_ = t_now_; } while (0)
// wave-level event counter of the counting build (-DD3IL_DEVICE_STATS -DD3IL_DEVICE_COUNTS; the atomics inside the contact loops distort the timers,
// so the plain stats build leaves them out): +1 per wave (its first active lane) each time the statement is reached
#if defined(D3IL_DEVICE_COUNTS)
#define PUSH_CNT(slot) do { if (__builtin_amdgcn_mbcnt_hi(__builtin_amdgcn_read_exec_hi(), __builtin_amdgcn_mbcnt_lo(__builtin_amdgcn_read_exec_lo(), 0u)) == 0 && PUSH_ROW_OK) \
      atomicAdd(&d3il::g_dev_cnt[PUSH_ROW][slot], 1ull); } while (0)
#else
#define PUSH_CNT(slot) ((void)0)
#endif
#else
#define PUSH_TIC ((void)0)
#define PUSH_TOC(slot) ((void)0)
#define PUSH_CNT(slot) ((void)0)
#endif

namespace d3il {
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

// flag bits (EnvState::flags).  F_TERMINATED / F_SUCCESS / F_IK_VALID / F_SOLVER_FAIL keep their Avoiding positions.
enum : unsigned {
  PF_FIRST_MASK = 0x7u,          // first_visit + 1   (pushing.py:341-377)
  PF_MODE_SHIFT = 3, PF_MODE_MASK = 0x7u << 3,   // mode + 1
  PF_WARM_VALID = 1u << 6,
  PF_CON_OVERFLOW = 1u << 18,    // more than PUSH_MAXCON contacts in one sub-step (extra contacts dropped)
  PF_OFF_TABLE = 1u << 19,       // a cube left the modelled part of the table top
};

struct BoxState { double pos[3], quat[4], vel[6]; };

// per-lane views of the scratch areas: h = the solver tables (LDS on the device), g = contact records (HBM), w = the environment's object block of the
// state buffer.  On the device the pointers carry their address space so that accesses compile to ds_* / global_* instructions instead of flat ones.
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(3))) double push_lds_double;
typedef __attribute__((address_space(1))) double push_glb_double;
#else
typedef double push_lds_double;
typedef double push_glb_double;
#endif
struct PushScratch {
  push_lds_double* h;
  push_glb_double* g; int gs;
  push_glb_double* w; int ws;
};
#define PWS(i) sc.w[(long)(i) * sc.ws]

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

#if defined(__HIPCC__)
#define D3IL_HOSTFN __host__
#else
#define D3IL_HOSTFN
#endif

}  // namespace d3il
