// gen_tree.h - phase 4 of the generic engine for islands whose contact graph is a TREE (included by gen_step.h).
//
// The constraint problem of a sub-step (MjScene.py:110-111 -> mj_step; solver options base.xml:3: Newton, elliptic cones) is
// block separable over the islands of {cubes, arm}.  Almost every island of the Sorting / Inserting scenes is a tree: a cube on
// its own, the rod on a cube, that cube pushing a second one, two cubes leaning on each other (tools/host/sort_island_hist.py).
// For a tree the Newton system has block-arrow structure and is solved EXACTLY by eliminating the cubes leaf to root:
//
//   * every cube is a node that lives in the REGISTERS of its own lane of the group (6 x 6 Hessian block, gradient, iterate);
//     the lane evaluates all contacts its cube takes part in (its own records and the cube <-> cube records of lower cubes that
//     name it - those are evaluated by both partners, nothing is accumulated across lanes);
//   * the arm hangs on the cube its rod touches as a 5-dimensional node "lambda": with W the arm rows of the rod contact (3) and
//     of the two finger limits (2), every Newton iterate that starts in a0 + range(M^-1 W') stays there, so the arm's nine
//     accelerations are x_a = a0 + M^-1 W' lambda and the arm block of the cost is 1/2 lambda' A lambda, A = W M^-1 W' -
//     constant over the iterations of a sub-step (lane 0 builds it once: gen_arm_reduce).  At the optimum lambda is the force;
//   * a child c of parent p sends the Schur complement  -H_pc H_cc^-1 [H_cp | g_c]  (21 + 6 doubles) through LDS; the root solves
//     its 6 x 6 (or lambda's 5 x 5) system; the directions travel back down the tree; the line search sums the lanes' partial
//     slopes through LDS slots.  Same iteration, line search and stopping rules as gen_solve (gen_step.h).
//
// Everything is unrolled over fixed sizes; the only memory traffic of an iteration are the contact records.  Islands that are not
// trees (three cubes touching each other, the rod on two cubes), an arm joint at its limit or rod <-> wall contacts of the arm
// block go through gen_solve as before.
// The host build runs the lanes of a group one after the other: every step below is a loop over the lane states (GT_FOR), one
// state on the device.
#pragma once

namespace d3il {


// Floating-point contraction in this file is the EXPRESSION-level rule (a * b + c written as one expression becomes an fma, nothing is fused across statements):
// gen_lone_solve and gen_tree_solve evaluate a cube on its own with the same expressions in the same order, and with this rule the same expressions become the
// same instructions in both - which of the two a wave runs may then depend on the other environments of the wave without an environment's result depending
// on them (tests/test_gpu_permutation.py, tools/gpu_perm_push_sort.py).  The helpers the two share are compiled here, under the same rule.
#if defined(__clang__)
#pragma clang fp contract(on)
#endif
D3IL_HD double gt_rsqrtd(double x) {   // rsqrtd (panda_step.h) with its two Newton steps written as explicit fused operations
#if defined(__HIP_DEVICE_COMPILE__)
  double y = __builtin_amdgcn_rsq(x);
  y = y * fma(-(0.5 * x * y), y, 1.5);
  y = y * fma(-(0.5 * x * y), y, 1.5);
  return y;
#else
  return 1.0 / sqrt(x);
#endif
}
D3IL_HD double gt_cone_eval(const double* jar, double Dn, double Dt, double mu, double fric, double* force, double* Hc /* 3x3 */) {
  if (Dn == 0) {   // inert row (inactive contact slot of this lane inside a wave-uniform loop)
#pragma unroll
    for (int i = 0; i < 9; i++) Hc[i] = 0;
    force[0] = force[1] = force[2] = 0;
    return 0;
  }
  double U0 = jar[0] * mu, U1 = jar[1] * fric, U2 = jar[2] * fric;
  double T2 = U1 * U1 + U2 * U2;
  double iT = T2 > 0 ? gt_rsqrtd(T2) : 0.0;
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

template <int N> D3IL_HD bool gt_ldl_n(double* A, double* d, double* id) {   // in place: strict lower part of A becomes L
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
template <int N> D3IL_HD void gt_ldl_solve_n(const double* L, const double* id, double* x) {
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

// LDS slots of the tree solver.  They alias the dense Hessian of gen_solve (GL_H), which runs after this phase.
constexpr int GT_MSG = GL_H;                          // per cube: Schur message to its parent, dH[21] dg[6]
constexpr int GT_A = GT_MSG + 27 * GEN_MAXNB;         // A = W M^-1 W', packed lower 5 x 5
constexpr int GT_CA = GT_A + 15;                      // W a0 (- aref for the finger rows)
constexpr int GT_DF = GT_CA + 5;                      // D of the finger rows (0: inert)
constexpr int GT_LAM = GT_DF + 2;                     // lambda
constexpr int GT_PL = GT_LAM + 5;                     // its Newton direction
constexpr int GT_HLL = GT_PL + 5;                     // lambda's own Hessian block (15) and gradient (5), then the original gradient (5)
constexpr int GT_EX = GT_HLL + 25;                    // exchange slots: 2 buffers x GEN_MAXNB lanes x 3
constexpr int GT_LS = GT_EX + 6 * GEN_MAXNB;           // line-search data of the first GT_LSCAP records of every cube: jar[3] jp[3] Dn fric (later records: fields 19 .. 27 of the record in HBM)
constexpr int GT_LSCAP = (GL_H + GEN_NH - GT_LS) / (8 * GEN_MAXNB);
constexpr int GT_END = GT_LS + 8 * GEN_MAXNB * GT_LSCAP;
static_assert(GT_END - GL_H <= GEN_NH && GT_LSCAP >= 8, "the tree solver's slots must fit the area they alias");
#define GT_LSS(b, q, k) GLS(GT_LS + 8 * (GT_LSCAP * (b) + (q)) + (k))
constexpr int GT_LAMNODE = GEN_MAXNB;                 // parent index that stands for the lambda node

struct GTLane {
  int b, cnt, parent, depth;
  unsigned members, children;
  bool fast, active, lam, failed;
  double x[6], g[6], g0[6], p[6], H[21], dd[6], idd[6], Hpc[36], fl5[5];
};

// is the island `mask` (bits of cubes) a tree the solver takes?  adj: symmetric cube adjacency, rodm: cubes under the rod
D3IL_HD bool gt_island_fast(const unsigned* adj, unsigned rodm, unsigned mask, int nb, bool red_ok) {
  int ncub = 0, deg = 0;
#pragma unroll
  for (int d = 0; d < GEN_MAXNB; d++) if (d < nb && ((mask >> d) & 1)) { ncub++; deg += __builtin_popcount(adj[d] & mask); }
  if (deg != 2 * (ncub - 1)) return false;
  if (rodm & mask) return red_ok;                       // the arm is part of it: only through the reduction (one rod contact, no arm joint at a limit)
  return true;
}
D3IL_HD void gt_graph(const GenConsts& gc_, const PushScratch sc, unsigned* adj, unsigned& rodm) {
  D3IL_GEN_CONSTS(gc_, gc);
  rodm = 0;
#pragma unroll
  for (int b = 0; b < GEN_MAXNB; b++) adj[b] = 0;
#pragma unroll
  for (int b = 0; b < GEN_MAXNB; b++) if (b < gc.nb) {
    const unsigned info = (unsigned)GLS(GL_INFO + b), m = (info >> 5) & 15u;
    adj[b] |= m;
#pragma unroll
    for (int d = 0; d < GEN_MAXNB; d++) if ((m >> d) & 1) adj[d] |= 1u << b;
    if ((info >> 9) & 1) rodm |= 1u << b;
  }
}
// the island of cube b and b's place in it
D3IL_HD void gt_analyse(const GenConsts& gc_, const PushScratch sc, int b, GTLane& t) {
  D3IL_GEN_CONSTS(gc_, gc);
  unsigned adj[GEN_MAXNB], rodm;
  gt_graph(gc, sc, adj, rodm);
  unsigned mask = 1u << b;
#pragma unroll
  for (int sweep = 0; sweep < GEN_MAXNB - 1; sweep++)
#pragma unroll
    for (int d = 0; d < GEN_MAXNB; d++) if ((mask >> d) & 1) mask |= adj[d];
  t.b = b; t.members = mask; t.cnt = (int)((unsigned)GLS(GL_INFO + b) & 31u);
  t.fast = gt_island_fast(adj, rodm, mask, gc.nb, GLS(GL_TR) == 1.0);
  t.failed = false; t.lam = false; t.parent = -1; t.depth = 0; t.children = 0;
  if (!t.fast) { t.active = false; return; }
  const bool arm_in = (rodm & mask) != 0;
  const int root = __builtin_ctz(arm_in ? (rodm & mask) : mask);
  int depth[GEN_MAXNB], parent[GEN_MAXNB];
#pragma unroll
  for (int d = 0; d < GEN_MAXNB; d++) { depth[d] = -1; parent[d] = -1; }
#pragma unroll
  for (int d = 0; d < GEN_MAXNB; d++) if (d == root) { depth[d] = arm_in ? 1 : 0; parent[d] = arm_in ? GT_LAMNODE : -1; }
#pragma unroll
  for (int sweep = 0; sweep < GEN_MAXNB - 1; sweep++) {
    const int dcur = (arm_in ? 1 : 0) + sweep;
#pragma unroll
    for (int c = 0; c < GEN_MAXNB; c++) if (depth[c] == dcur)
#pragma unroll
      for (int n = 0; n < GEN_MAXNB; n++) if (((adj[c] & mask) >> n) & 1) if (depth[n] < 0) { depth[n] = dcur + 1; parent[n] = c; }
  }
#pragma unroll
  for (int d = 0; d < GEN_MAXNB; d++) {
    if (d == b) { t.depth = depth[d]; t.parent = parent[d]; }
    if (parent[d] == b) t.children |= 1u << d;
  }
  t.lam = arm_in && b == root;
  t.active = t.cnt > 0 || t.members != (1u << b);      // a cube without any contact: x = a0
}

// ---- lane 0, after phase 3b: the arm's reduction to the lambda node (exactly one rod <-> cube contact, no arm joint at a limit, no contact of
// the arm block itself).  Publishes A, W a0, the finger rows' D and the warm start lambda0 = A^-1 W (x_warm - a0), which reproduces the warm
// start's row residuals with the least smooth cost.  GL_TR = 1 when the reduction stands.
template <bool RS>
D3IL_NOINLINE inline void gen_arm_reduce(const GenConsts& gc_, const PushScratch sc, bool warm_valid) {
  D3IL_GEN_CONSTS(gc_, gc);
  const int arm0 = 6 * gc.nb;
  GLS(GL_TR) = 0.0; GLS(GT_HLL + 24) = 0.0;
  int rb = -1, nrod = 0;
#pragma unroll
  for (int b = 0; b < GEN_MAXNB; b++) if (b < gc.nb && (((unsigned)GLS(GL_INFO + b) >> 9) & 1)) { rb = b; nrod++; }
  if (nrod != 1 || GLS(GL_INFO + 4) != 0) return;
  if (RS && GLS(GL_INFO + 8) != 0) return;
  double M[45], L[45], d[NDOF], id[NDOF];
#pragma unroll
  for (int i = 0; i < 45; i++) M[i] = GLS(GL_M + i);
  if (!ldl9(M, L, d, id)) return;
  double sf[NFING], a0[NDOF];
#pragma unroll
  for (int k = 0; k < NDOF; k++) a0[k] = GLS(GL_A0 + arm0 + k);
#pragma unroll
  for (int f = 0; f < NFING; f++) {
    const double s = GLS(GL_LIM + 3 * (NARM + f));
    sf[f] = s != 0 ? s : 1.0;
    GLS(GT_DF + f) = s != 0 ? GLS(GL_LIM + 3 * (NARM + f) + 1) : 0.0;
    GLS(GT_CA + 3 + f) = sf[f] * a0[NARM + f] - (s != 0 ? GLS(GL_LIM + 3 * (NARM + f) + 2) : 0.0);
  }
  double A[15], wu[5];
#pragma unroll
  for (int s = 0; s < 5; s++) {                   // column s of Y = M^-1 W', then the entries A(r, s), r >= s
    double y[NDOF];
#pragma unroll
    for (int k = 0; k < NDOF; k++) y[k] = 0;
    if (s < 3) {
#pragma unroll
      for (int k = 0; k < NARM; k++) y[k] = GLS(GL_JA + 21 * rb + 7 * s + k);
    } else y[NARM + s - 3] = sf[s - 3];
    ldl9_solve(L, id, y);
#pragma unroll
    for (int r = 0; r < 5; r++) if (r >= s) {
      double acc = 0;
      if (r < 3) {
#pragma unroll
        for (int k = 0; k < NARM; k++) acc += GLS(GL_JA + 21 * rb + 7 * r + k) * y[k];
      } else acc = sf[r - 3] * y[NARM + r - 3];
      A[tri(r, s)] = acc;
    }
  }
#pragma unroll
  for (int r = 0; r < 3; r++) {
    double acc = 0, au = 0;
#pragma unroll
    for (int k = 0; k < NARM; k++) { const double j = GLS(GL_JA + 21 * rb + 7 * r + k); acc += j * a0[k]; au += j * (warm_valid ? GWARM(arm0 + k) - a0[k] : 0.0); }
    GLS(GT_CA + r) = acc; wu[r] = au;
  }
#pragma unroll
  for (int f = 0; f < NFING; f++) wu[3 + f] = warm_valid ? sf[f] * (GWARM(arm0 + NARM + f) - a0[NARM + f]) : 0.0;
#pragma unroll
  for (int i = 0; i < 15; i++) GLS(GT_A + i) = A[i];
  double d5[5], id5[5];
  if (!ldl_n<5>(A, d5, id5)) return;
  ldl_solve_n<5>(A, id5, wu);
#pragma unroll
  for (int r = 0; r < 5; r++) GLS(GT_LAM + r) = wu[r];
  GLS(GL_TR) = 1.0;
}

// ---- contact terms in point form.  Inside the solve a cube's six unknowns are its linear acceleration and its angular acceleration in WORLD axes
// (the cubes' inertia is isotropic - build_gen_consts checks it -, so the smooth block m I | i I is the same in both frames; the iterate is turned
// into the body axes of the free joint once, when the solve ends).  With r = contact position - cube centre the acceleration of the cube's material
// point is  G y = y_lin + y_ang x r, G = [I, -[r]x], and a contact row is  sign f' G: everything a contact contributes follows from 3-vectors -
//   jar = F u - aref,  u = sum over the two bodies of sign G y;    w = F' force,  S = F' Hc F  (world 3 x 3);
//   gradient  -= sign [w, r x w];    diagonal block += G' S G = [S, -P; -P', Q],  P = S [r]x,  Q = -[r]x P;
//   block between two bodies (signs multiply to -1)  = -[S, -Pc; -Pp', -[rp]x Pc]
// - about a quarter of the multiply-adds of the 3 x 6 row form J' Hc J.
D3IL_HD void gt_point(const double* y, const double* r, double sgn, double* u) {       // u += sgn (y_lin + y_ang x r)
  u[0] += sgn * (y[0] + y[4] * r[2] - y[5] * r[1]);
  u[1] += sgn * (y[1] + y[5] * r[0] - y[3] * r[2]);
  u[2] += sgn * (y[2] + y[3] * r[1] - y[4] * r[0]);
}
D3IL_HD void gt_point_lds(const PushScratch sc, int off, const double* r, double sgn, double* u) {     // the same with y in the t area
  double y[6];
#pragma unroll
  for (int k = 0; k < 6; k++) y[k] = GLS(off + k);
  gt_point(y, r, sgn, u);
}
D3IL_HD void gt_world_S(const double* F, const double* Hc, double* S) {      // S = F' Hc F, full 3 x 3 (symmetric)
  double T[9];
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int j = 0; j < 3; j++) T[3 * a + j] = Hc[3 * a] * F[j] + Hc[3 * a + 1] * F[3 + j] + Hc[3 * a + 2] * F[6 + j];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = i; j < 3; j++) { const double s = F[i] * T[j] + F[3 + i] * T[3 + j] + F[6 + i] * T[6 + j]; S[3 * i + j] = s; S[3 * j + i] = s; }
}
D3IL_HD void gt_SxR(const double* S, const double* r, double* P) {            // P = S [r]x
#pragma unroll
  for (int i = 0; i < 3; i++) {
    P[3 * i] = S[3 * i + 1] * r[2] - S[3 * i + 2] * r[1];
    P[3 * i + 1] = S[3 * i + 2] * r[0] - S[3 * i] * r[2];
    P[3 * i + 2] = S[3 * i] * r[1] - S[3 * i + 1] * r[0];
  }
}
D3IL_HD void gt_add_diag(double* H, const double* S, const double* P, const double* r) {      // packed lower 6 x 6 += G' S G
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j <= i; j++) H[tri(i, j)] += S[3 * i + j];
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int l = 0; l < 3; l++) H[tri(3 + a, l)] -= P[3 * l + a];
  // Q(:, j) = -(r x P(:, j)), lower part
  H[tri(3, 3)] -= r[1] * P[6] - r[2] * P[3];
  H[tri(4, 3)] -= r[2] * P[0] - r[0] * P[6];
  H[tri(5, 3)] -= r[0] * P[3] - r[1] * P[0];
  H[tri(4, 4)] -= r[2] * P[1] - r[0] * P[7];
  H[tri(5, 4)] -= r[0] * P[4] - r[1] * P[1];
  H[tri(5, 5)] -= r[0] * P[5] - r[1] * P[2];
}
// X(j, i), rows: the parent's unknowns j, columns: this cube's i, += sg Gp' S Gc  (sg = product of the two signs); Pc = S [rc]x
D3IL_HD void gt_add_off(double* X, const double* S, const double* Pc, const double* rp, double sg) {
  double Pp[9];
  gt_SxR(S, rp, Pp);
#pragma unroll
  for (int l = 0; l < 3; l++) {
#pragma unroll
    for (int m = 0; m < 3; m++) { X[6 * l + m] += sg * S[3 * l + m]; X[6 * l + 3 + m] -= sg * Pc[3 * l + m]; X[6 * (3 + l) + m] -= sg * Pp[3 * m + l]; }
  }
#pragma unroll
  for (int b = 0; b < 3; b++) {      // -(rp x Pc(:, b))
    X[6 * 3 + 3 + b] -= sg * (rp[1] * Pc[6 + b] - rp[2] * Pc[3 + b]);
    X[6 * 4 + 3 + b] -= sg * (rp[2] * Pc[b] - rp[0] * Pc[6 + b]);
    X[6 * 5 + 3 + b] -= sg * (rp[0] * Pc[3 + b] - rp[1] * Pc[b]);
  }
}
D3IL_HD double gt_symdot5(const double* A, const double* u, const double* v) {     // u' A v, A packed lower 5 x 5
  double s = 0;
#pragma unroll
  for (int r = 0; r < 5; r++)
#pragma unroll
    for (int c = 0; c < 5; c++) s += u[r] * A[r >= c ? tri(r, c) : tri(c, r)] * v[c];
  return s;
}
// world axes <-> body axes of cube b's angular unknowns (R row-major, columns = body axes)
D3IL_HD void gt_to_world(const PushScratch sc, int b, double* y) {
  double R[9];
#pragma unroll
  for (int k = 0; k < 9; k++) R[k] = GLS(GL_R + 9 * b + k);
  const double a = y[3], bb = y[4], c = y[5];
  y[3] = R[0] * a + R[1] * bb + R[2] * c; y[4] = R[3] * a + R[4] * bb + R[5] * c; y[5] = R[6] * a + R[7] * bb + R[8] * c;
}
D3IL_HD void gt_to_body(const PushScratch sc, int b, double* y) {
  double R[9];
#pragma unroll
  for (int k = 0; k < 9; k++) R[k] = GLS(GL_R + 9 * b + k);
  const double a = y[3], bb = y[4], c = y[5];
  y[3] = R[0] * a + R[3] * bb + R[6] * c; y[4] = R[1] * a + R[4] * bb + R[7] * c; y[5] = R[2] * a + R[5] * bb + R[8] * c;
}

// v + the value of the pair's other sub-lane (device: DPP swap of neighbouring lanes - quad_perm [1, 0, 3, 2] -, the same sum in both lanes since
// the addition commutes; host: one sub-lane, nothing to add).  The wave barriers keep the optimiser from moving code across the cross-lane
// operation (common-code sinking around convergent operations).
D3IL_HD double gt_pair_sum(double v) {
#if defined(__HIP_DEVICE_COMPILE__) && D3IL_GEN_NSUB == 2
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int plo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, false), phi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, false);
  return v + __hiloint2double(phi, plo);
#else
  return v;
#endif
}
template <int N> D3IL_HD void gt_pair_sum_n(double* v) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int k = 0; k < N; k++) v[k] = gt_pair_sum(v[k]);
  __builtin_amdgcn_wave_barrier();
#else
  (void)v;
#endif
}

// The solver as a function of its own saves and restores ~160 callee-saved registers around every call - 84 KB of scratch stores per wave and sub-step, which
// this L2 does not keep (13 x the step's write traffic, profiles/r05/README.md).  -DD3IL_GT_INLINE builds it into its caller instead (A/B).
#if defined(D3IL_GT_INLINE)
#define D3IL_GT_SOLVE_ATTR D3IL_HD
#elif defined(D3IL_GT_STATIC)
#define D3IL_GT_SOLVE_ATTR static D3IL_NOINLINE      // internal linkage: inter-procedural register allocation may drop the callee-saved convention for it
#else
#define D3IL_GT_SOLVE_ATTR D3IL_NOINLINE
#endif
#define GT_FOR(li) for (int li = 0; li < NL; li++)

// sum / max over the members of the lane's island of what the lanes put into exchange buffer `buf`
#define GT_PUT(t, buf, k, v) GLS(GT_EX + 3 * GEN_MAXNB * (buf) + 3 * (t).b + (k)) = (v)
#define GT_GET(buf, c, k) GLS(GT_EX + 3 * GEN_MAXNB * (buf) + 3 * (c) + (k))

// Newton solve of the tree islands of the environment by the lanes of cubes l0 .. l0 + NL - 1 (device: NL = 1, this lane's cube; host and the
// one-lane reset kernel: all cubes, one after the other in every step).  Returns the flags it raises.
// NS, sub: sub-lanes per cube and the lane's place in its pair (step kernel: NS = GEN_NSUB = 2; everywhere else one lane per cube).  The sub-lanes of a pair run this function with identical data and take
// identical decisions; only the contact loops differ - sub-lane s evaluates records s, s + NS, .. - and their sums are exchanged (gt_pair_sum).
template <int NL, int NS = 1>
D3IL_GT_SOLVE_ATTR inline unsigned gen_tree_solve(const GenConsts& gc_, const PushScratch sc, int l0, bool warm_valid, int sub = 0) {
  D3IL_GEN_CONSTS(gc_, gc);
  unsigned fl = 0;
  GTLane t[NL];
  GT_FOR(li) {
    GTLane& T = t[li];
    T.b = l0 + li; T.fast = false; T.active = false; T.lam = false; T.failed = false; T.cnt = 0; T.members = 0; T.children = 0; T.parent = -1; T.depth = 0;
    if (l0 + li < gc.nb) gt_analyse(gc, sc, l0 + li, T);
  }
  const int arm0 = 6 * gc.nb;
  const double impr = gc.impratio, mu_scale = sqrt(1 / fmax(1e-15, impr)), mt = gc.box_mass, mr = gc.box_inertia;
  const double grav2 = GLS(GL_A0 + 2);                 // every cube's smooth acceleration is gravity (gen_phase2)
  PUSH_TIC;
  // ---- start point (world axes, see above); the cubes' velocities in world axes are published at GL_P for the partners' reference accelerations
  GT_FOR(li) {
    GTLane& T = t[li];
#pragma unroll
    for (int k = 0; k < 6; k++) { T.x[k] = 0; T.p[k] = 0; }
    if (!T.fast) continue;
#pragma unroll
    for (int k = 0; k < 6; k++) T.x[k] = warm_valid ? GWARM(6 * T.b + k) : GLS(GL_A0 + 6 * T.b + k);
    if (!T.active) {
#pragma unroll
      for (int k = 0; k < 6; k++) GLS(GL_X + 6 * T.b + k) = GLS(GL_A0 + 6 * T.b + k);
      continue;
    }
    gt_to_world(sc, T.b, T.x);
#pragma unroll
    for (int k = 0; k < 6; k++) GLS(GL_X + 6 * T.b + k) = T.x[k];
  }
  GT_FOR(li) {
    GTLane& T = t[li];
    if (l0 + li >= gc.nb) continue;
    double vel[6];
#pragma unroll
    for (int k = 0; k < 6; k++) vel[k] = GLS(GL_VEL + 6 * T.b + k);
    gt_to_world(sc, T.b, vel);
#pragma unroll
    for (int k = 0; k < 6; k++) GLS(GL_P + 6 * T.b + k) = vel[k];
  }
  gen_sync();
  // ---- reference acceleration and regularisation of the lane's own records
  GT_FOR(li) {
    GTLane& T = t[li];
    if (!T.active) continue;
    double vel[6], pc[3];
#pragma unroll
    for (int k = 0; k < 6; k++) vel[k] = GLS(GL_P + 6 * T.b + k);
#pragma unroll
    for (int k = 0; k < 3; k++) pc[k] = GLS(GL_POS + 3 * T.b + k);
#pragma clang loop unroll(disable)
    for (int q = sub; q < T.cnt; q += NS) {
      const int base = GG_CON + (T.b * GEN_SEG + q) * GREC;
      double rec[16];
#pragma unroll
      for (int k = 0; k < 16; k++) rec[k] = GRS(base + k);
      const int kind = (int)rec[13], a = (int)rec[14], bb = (int)rec[15];
      const int set = kind == GK_STATIC ? a : (kind == GK_BOXBOX ? gc.set_bb : gc.set_rod);
      const double invw = kind == GK_STATIC ? gc.box_invw_t : (kind == GK_BOXBOX ? 2 * gc.box_invw_t : gc.box_invw_t + gc.rod_invw);
      const double r1[3] = {rec[0] - pc[0], rec[1] - pc[1], rec[2] - pc[2]};
      double u[3] = {0, 0, 0}, v[3];
      gt_point(vel, r1, kind == GK_STATIC ? (gc.st_first[a] ? 1.0 : -1.0) : -1.0, u);
      if (kind == GK_BOXBOX) {
        const double r2[3] = {rec[0] - GLS(GL_POS + 3 * bb), rec[1] - GLS(GL_POS + 3 * bb + 1), rec[2] - GLS(GL_POS + 3 * bb + 2)};
        gt_point_lds(sc, GL_P + 6 * bb, r2, 1.0, u);
      }
#pragma unroll
      for (int r = 0; r < 3; r++) v[r] = rec[3 + 3 * r] * u[0] + rec[4 + 3 * r] * u[1] + rec[5 + 3 * r] * u[2];
      if (kind == GK_ROD) {
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
          for (int k = 0; k < NARM; k++) v[r] += GLS(GL_JA + 21 * T.b + 7 * r + k) * GLS(GL_VEL + arm0 + k);
      }
      const double dist = rec[12], imp = impedance(gc.ct_solimp[set], dist);
      GRS(base + 16) = -gc.ct_B[set] * v[0] - gc.ct_K[set] * imp * dist;
      GRS(base + 17) = -gc.ct_B[set] * v[1]; GRS(base + 18) = -gc.ct_B[set] * v[2];
      const double Dn = 1 / fmax(1e-15, (1 - imp) / imp * invw);
      GRS(base + 19) = Dn;
      if (q < GT_LSCAP) { GT_LSS(T.b, q, 6) = Dn; GT_LSS(T.b, q, 7) = gc.ct_fric[set]; }
    }
    D3IL_STAT(g_stats.newton_calls++);
  }
  gen_sync();
  PUSH_TOC(3);
  int buf = 0;
  bool any = false;
  GT_FOR(li) any = any || t[li].active;
  if (any) PUSH_CNT(6);
#pragma clang loop unroll(disable)
  for (int it = 0; it < 60 && any; it++) {
    // ---- gradient and Hessian blocks at x: smooth part, own records, the cube <-> cube records of lower partners, the lambda node
    GT_FOR(li) {
      GTLane& T = t[li];
      if (!T.active) continue;
      D3IL_STAT(g_stats.newton_iters++);
      PUSH_CNT(0);
#pragma unroll
      for (int i = 0; i < 21; i++) T.H[i] = 0;
#pragma unroll
      for (int i = 0; i < 36; i++) T.Hpc[i] = 0;
#pragma unroll
      for (int k = 0; k < 6; k++) { const double mm = sub == 0 ? (k < 3 ? mt : mr) : 0.0; T.g[k] = mm * (T.x[k] - (k == 2 ? grav2 : 0.0)); T.H[tri(k, k)] = mm; }      // the smooth part once per pair
#pragma unroll
      for (int k = 0; k < 5; k++) T.fl5[k] = 0;
      double pc[3];
#pragma unroll
      for (int k = 0; k < 3; k++) pc[k] = GLS(GL_POS + 3 * T.b + k);
#pragma clang loop unroll(disable)
      for (int q = sub; q < T.cnt; q += NS) {
        const int base = GG_CON + (T.b * GEN_SEG + q) * GREC;
        double rec[22];
#pragma unroll
        for (int k = 0; k < 22; k++) rec[k] = GRS(base + k);
        const int kind = (int)rec[13], bb = (int)rec[15];
        PUSH_CNT(1);
        const double sg = rec[21];
        const double r1[3] = {rec[0] - pc[0], rec[1] - pc[1], rec[2] - pc[2]};
        double r2[3] = {0, 0, 0}, u[3] = {0, 0, 0}, jar[3], force[3], Hc[9], B[3][5];
        gt_point(T.x, r1, sg, u);
        bool off = false;
        if (kind == GK_BOXBOX) {
#pragma unroll
          for (int k = 0; k < 3; k++) r2[k] = rec[k] - GLS(GL_POS + 3 * bb + k);
          gt_point_lds(sc, GL_X + 6 * bb, r2, 1.0, u);
          off = bb == T.parent;
        }
#pragma unroll
        for (int r = 0; r < 3; r++) jar[r] = rec[3 + 3 * r] * u[0] + rec[4 + 3 * r] * u[1] + rec[5 + 3 * r] * u[2] - rec[16 + r];
        if (kind == GK_ROD) {      // the arm through the lambda node: J_a x_a = A(r, :) lambda + J_a a0
#pragma unroll
          for (int r = 0; r < 3; r++) {
            double s = GLS(GT_CA + r);
#pragma unroll
            for (int k = 0; k < 5; k++) { const double ak = GLS(GT_A + (r >= k ? tri(r, k) : tri(k, r))); B[r][k] = ak; s += ak * GLS(GT_LAM + k); }
            jar[r] += s;
          }
        }
        if (q < GT_LSCAP) {
#pragma unroll
          for (int r = 0; r < 3; r++) GT_LSS(T.b, q, r) = jar[r];
        } else {
#pragma unroll
          for (int r = 0; r < 3; r++) GRS(base + 22 + r) = jar[r];
        }
        const double Dn = rec[19], fric = rec[20];
        gt_cone_eval(jar, Dn, Dn * impr, fric * mu_scale, fric, force, Hc);
        if (force[0] == 0 && force[1] == 0 && force[2] == 0) continue;
        double w[3], S[9], P[9];
#pragma unroll
        for (int k = 0; k < 3; k++) w[k] = sg * (rec[3 + k] * force[0] + rec[6 + k] * force[1] + rec[9 + k] * force[2]);
        T.g[0] -= w[0]; T.g[1] -= w[1]; T.g[2] -= w[2];
        T.g[3] -= r1[1] * w[2] - r1[2] * w[1]; T.g[4] -= r1[2] * w[0] - r1[0] * w[2]; T.g[5] -= r1[0] * w[1] - r1[1] * w[0];
        gt_world_S(rec + 3, Hc, S);
        gt_SxR(S, r1, P);
        gt_add_diag(T.H, S, P, r1);
        if (off) gt_add_off(T.Hpc, S, P, r2, -1.0);
        if (kind == GK_ROD) {             // lambda's own block from this contact: A_r' Hc A_r, the block between lambda and this cube, the force for lambda's gradient
#pragma unroll
          for (int r = 0; r < 3; r++) T.fl5[r] = force[r];
#pragma unroll
          for (int j = 0; j < 5; j++) {
            double tb[3], wv[3];
#pragma unroll
            for (int s = 0; s < 3; s++) tb[s] = B[0][j] * Hc[s] + B[1][j] * Hc[3 + s] + B[2][j] * Hc[6 + s];
#pragma unroll
            for (int i = 0; i <= j; i++) GLS(GT_HLL + tri(j, i)) = tb[0] * B[0][i] + tb[1] * B[1][i] + tb[2] * B[2][i];
#pragma unroll
            for (int k = 0; k < 3; k++) wv[k] = sg * (tb[0] * rec[3 + k] + tb[1] * rec[6 + k] + tb[2] * rec[9 + k]);
            T.Hpc[6 * j] += wv[0]; T.Hpc[6 * j + 1] += wv[1]; T.Hpc[6 * j + 2] += wv[2];
            T.Hpc[6 * j + 3] += r1[1] * wv[2] - r1[2] * wv[1]; T.Hpc[6 * j + 4] += r1[2] * wv[0] - r1[0] * wv[2]; T.Hpc[6 * j + 5] += r1[0] * wv[1] - r1[1] * wv[0];
          }
          GLS(GT_HLL + 24) = 1.0;         // marks "the rod contact is active" for the assembly below
        }
      }
      // cube <-> cube records held by lower partners (this cube is their geom 2)
#pragma unroll
      for (int c = 0; c < GEN_MAXNB; c++) if (c < T.b && (((unsigned)GLS(GL_INFO + c) >> (5 + T.b)) & 1)) {
        const unsigned pr = (unsigned)GLS(GL_PAIR + gt_pair(c, T.b));
        const int q0 = (int)(pr & 31u), q1 = q0 + (int)(pr >> 5);
        double pp[3], xc[6];
#pragma unroll
        for (int k = 0; k < 3; k++) pp[k] = GLS(GL_POS + 3 * c + k);
#pragma unroll
        for (int k = 0; k < 6; k++) xc[k] = GLS(GL_X + 6 * c + k);
#pragma clang loop unroll(disable)
        for (int q = q0 + sub; q < q1; q += NS) {
          const int base = GG_CON + (c * GEN_SEG + q) * GREC;
          double rec[21];
#pragma unroll
          for (int k = 0; k < 21; k++) rec[k] = GRS(base + k);
          const double r1[3] = {rec[0] - pc[0], rec[1] - pc[1], rec[2] - pc[2]}, r2[3] = {rec[0] - pp[0], rec[1] - pp[1], rec[2] - pp[2]};
          PUSH_CNT(2);
          double u[3] = {0, 0, 0}, jar[3], force[3], Hc[9];
          gt_point(T.x, r1, 1.0, u);
          gt_point(xc, r2, -1.0, u);
#pragma unroll
          for (int r = 0; r < 3; r++) jar[r] = rec[3 + 3 * r] * u[0] + rec[4 + 3 * r] * u[1] + rec[5 + 3 * r] * u[2] - rec[16 + r];
          const double Dn = rec[19], fric = rec[20];
          gt_cone_eval(jar, Dn, Dn * impr, fric * mu_scale, fric, force, Hc);
          if (force[0] == 0 && force[1] == 0 && force[2] == 0) continue;
          double w[3], S[9], P[9];
#pragma unroll
          for (int k = 0; k < 3; k++) w[k] = rec[3 + k] * force[0] + rec[6 + k] * force[1] + rec[9 + k] * force[2];
          T.g[0] -= w[0]; T.g[1] -= w[1]; T.g[2] -= w[2];
          T.g[3] -= r1[1] * w[2] - r1[2] * w[1]; T.g[4] -= r1[2] * w[0] - r1[0] * w[2]; T.g[5] -= r1[0] * w[1] - r1[1] * w[0];
          gt_world_S(rec + 3, Hc, S);
          gt_SxR(S, r1, P);
          gt_add_diag(T.H, S, P, r1);
          if (c == T.parent) gt_add_off(T.Hpc, S, P, r2, -1.0);
        }
      }
      if (NS > 1) {      // the pair's sums; the rod contact's lambda block and marker were stored by the sub-lane that holds the contact
        gt_pair_sum_n<6>(T.g); gt_pair_sum_n<21>(T.H); gt_pair_sum_n<36>(T.Hpc); gt_pair_sum_n<3>(T.fl5);
        gen_sync();
      }
#pragma unroll
      for (int k = 0; k < 6; k++) T.g0[k] = T.g[k];
      double gm = 0;
#pragma unroll
      for (int k = 0; k < 6; k++) gm = fmax(gm, fabs(T.g[k]));
      if (T.lam) {      // lambda's block: A + (rod part, stored above when the contact is active) + finger rows; gradient A (lambda - f)
        double lam[5], A[15], dl[5];
        const bool rod_on = GLS(GT_HLL + 24) == 1.0;
        GLS(GT_HLL + 24) = 0.0;
#pragma unroll
        for (int i = 0; i < 15; i++) A[i] = GLS(GT_A + i);
#pragma unroll
        for (int k = 0; k < 5; k++) lam[k] = GLS(GT_LAM + k);
        double Hll[15];
#pragma unroll
        for (int i = 0; i < 15; i++) Hll[i] = A[i] + (rod_on ? GLS(GT_HLL + i) : 0.0);
#pragma unroll
        for (int f = 0; f < NFING; f++) {
          const double D = GLS(GT_DF + f);
          double jf = GLS(GT_CA + 3 + f);
#pragma unroll
          for (int k = 0; k < 5; k++) jf += A[3 + f >= k ? tri(3 + f, k) : tri(k, 3 + f)] * lam[k];
          if (D > 0 && jf < 0) {
            T.fl5[3 + f] = -D * jf;
#pragma unroll
            for (int j = 0; j < 5; j++)
#pragma unroll
              for (int i = 0; i <= j; i++) Hll[tri(j, i)] += D * A[3 + f >= j ? tri(3 + f, j) : tri(j, 3 + f)] * A[3 + f >= i ? tri(3 + f, i) : tri(i, 3 + f)];
          }
        }
#pragma unroll
        for (int k = 0; k < 5; k++) dl[k] = lam[k] - T.fl5[k];
#pragma unroll
        for (int r = 0; r < 5; r++) { double s = 0;
#pragma unroll
          for (int c = 0; c < 5; c++) s += A[r >= c ? tri(r, c) : tri(c, r)] * dl[c];
          GLS(GT_HLL + 15 + r) = s; GLS(GT_HLL + 20 + r) = s; }
#pragma unroll
        for (int i = 0; i < 15; i++) GLS(GT_HLL + i) = Hll[i];
        // the gradient in the arm's own coordinates, W' (lambda - f), for the stopping rule
#pragma unroll
        for (int k = 0; k < NARM; k++) gm = fmax(gm, fabs(GLS(GL_JA + 21 * T.b + k) * dl[0] + GLS(GL_JA + 21 * T.b + 7 + k) * dl[1] + GLS(GL_JA + 21 * T.b + 14 + k) * dl[2]));
        gm = fmax(gm, fmax(fabs(dl[3]), fabs(dl[4])));
      }
      GT_PUT(T, buf, 0, gm);
    }
    gen_sync();
    GT_FOR(li) {
      GTLane& T = t[li];
      if (!T.active) continue;
      double gm = 0;
#pragma unroll
      for (int c = 0; c < GEN_MAXNB; c++) if ((T.members >> c) & 1) gm = fmax(gm, GT_GET(buf, c, 0));
      if (gm <= PUSH_GRAD_TOL) T.active = false;
    }
    buf ^= 1;
    PUSH_TOC(4);
    // ---- elimination, leaves first: a cube at depth d factorises its block and sends the Schur complement to its parent
#pragma clang loop unroll(disable)
    for (int d = GEN_MAXNB; d >= 1; d--) {
      GT_FOR(li) {
        GTLane& T = t[li];
        if (!T.active || T.depth != d) continue;
        if (!gt_ldl_n<6>(T.H, T.dd, T.idd)) T.failed = true;
        double v[6], dH[21], dg[6];
#pragma unroll
        for (int k = 0; k < 6; k++) v[k] = T.g[k];
        gt_ldl_solve_n<6>(T.H, T.idd, v);
#pragma unroll
        for (int j = 0; j < 6; j++) { double s = 0;
#pragma unroll
          for (int i = 0; i < 6; i++) s += T.Hpc[6 * j + i] * v[i];
          dg[j] = -s; }
#pragma unroll
        for (int k = 0; k < 6; k++) {
          double w[6];
#pragma unroll
          for (int i = 0; i < 6; i++) w[i] = T.Hpc[6 * k + i];
          gt_ldl_solve_n<6>(T.H, T.idd, w);
#pragma unroll
          for (int j = k; j < 6; j++) { double s = 0;
#pragma unroll
            for (int i = 0; i < 6; i++) s += T.Hpc[6 * j + i] * w[i];
            dH[tri(j, k)] = -s; }
        }
        if (T.parent == GT_LAMNODE) {      // the lambda node lives on this lane: its system, then this cube's direction
          double Hll[15], gl[5], d5[5], id5[5];
#pragma unroll
          for (int j = 0; j < 5; j++)
#pragma unroll
            for (int i = 0; i <= j; i++) Hll[tri(j, i)] = GLS(GT_HLL + tri(j, i)) + dH[tri(j, i)];
#pragma unroll
          for (int j = 0; j < 5; j++) gl[j] = -(GLS(GT_HLL + 15 + j) + dg[j]);
          if (!ldl_n<5>(Hll, d5, id5)) T.failed = true;
          ldl_solve_n<5>(Hll, id5, gl);
#pragma unroll
          for (int j = 0; j < 5; j++) GLS(GT_PL + j) = gl[j];
#pragma unroll
          for (int i = 0; i < 6; i++) { double s = -T.g[i];
#pragma unroll
            for (int j = 0; j < 5; j++) s -= T.Hpc[6 * j + i] * gl[j];
            T.p[i] = s; }
          gt_ldl_solve_n<6>(T.H, T.idd, T.p);
#pragma unroll
          for (int k = 0; k < 6; k++) GLS(GL_P + 6 * T.b + k) = T.p[k];
        } else {
#pragma unroll
          for (int i = 0; i < 21; i++) GLS(GT_MSG + 27 * T.b + i) = dH[i];
#pragma unroll
          for (int i = 0; i < 6; i++) GLS(GT_MSG + 27 * T.b + 21 + i) = dg[i];
        }
      }
      gen_sync();
      GT_FOR(li) {
        GTLane& T = t[li];
        if (!T.active || T.depth != d - 1 || T.children == 0) continue;
#pragma unroll
        for (int c = 0; c < GEN_MAXNB; c++) if ((T.children >> c) & 1) {
#pragma unroll
          for (int i = 0; i < 21; i++) T.H[i] += GLS(GT_MSG + 27 * c + i);
#pragma unroll
          for (int i = 0; i < 6; i++) T.g[i] += GLS(GT_MSG + 27 * c + 21 + i);
        }
      }
    }
    // ---- roots that are cubes; then the directions travel down the tree
    GT_FOR(li) {
      GTLane& T = t[li];
      if (!T.active || T.depth != 0) continue;
      if (!gt_ldl_n<6>(T.H, T.dd, T.idd)) T.failed = true;
#pragma unroll
      for (int k = 0; k < 6; k++) T.p[k] = -T.g[k];
      gt_ldl_solve_n<6>(T.H, T.idd, T.p);
#pragma unroll
      for (int k = 0; k < 6; k++) GLS(GL_P + 6 * T.b + k) = T.p[k];
    }
    gen_sync();
#pragma clang loop unroll(disable)
    for (int d = 1; d <= GEN_MAXNB; d++) {
      GT_FOR(li) {
        GTLane& T = t[li];
        if (!T.active || T.depth != d || T.parent == GT_LAMNODE) continue;
        double pp[6];
#pragma unroll
        for (int j = 0; j < 6; j++) pp[j] = GLS(GL_P + 6 * T.parent + j);
#pragma unroll
        for (int i = 0; i < 6; i++) { double s = -T.g[i];
#pragma unroll
          for (int j = 0; j < 6; j++) s -= T.Hpc[6 * j + i] * pp[j];
          T.p[i] = s; }
        gt_ldl_solve_n<6>(T.H, T.idd, T.p);
#pragma unroll
        for (int k = 0; k < 6; k++) GLS(GL_P + 6 * T.b + k) = T.p[k];
      }
      gen_sync();
    }
    PUSH_TOC(5);
    // ---- directional derivatives of the own records; the smooth terms of the line search
    GT_FOR(li) {
      GTLane& T = t[li];
      if (!T.active) continue;
      double pc[3];
#pragma unroll
      for (int k = 0; k < 3; k++) pc[k] = GLS(GL_POS + 3 * T.b + k);
#pragma clang loop unroll(disable)
      for (int q = sub; q < T.cnt; q += NS) {
        const int base = GG_CON + (T.b * GEN_SEG + q) * GREC;
        double rec[15];      // pos[3] frame[9] kind b sign
#pragma unroll
        for (int k = 0; k < 12; k++) rec[k] = GRS(base + k);
        rec[12] = GRS(base + 13); rec[13] = GRS(base + 15); rec[14] = GRS(base + 21);
        const int kind = (int)rec[12], bb = (int)rec[13];
        const double r1[3] = {rec[0] - pc[0], rec[1] - pc[1], rec[2] - pc[2]};
        PUSH_CNT(5);
        double u[3] = {0, 0, 0}, jp[3];
        gt_point(T.p, r1, rec[14], u);
        if (kind == GK_BOXBOX) {
          const double r2[3] = {rec[0] - GLS(GL_POS + 3 * bb), rec[1] - GLS(GL_POS + 3 * bb + 1), rec[2] - GLS(GL_POS + 3 * bb + 2)};
          gt_point_lds(sc, GL_P + 6 * bb, r2, 1.0, u);
        }
#pragma unroll
        for (int r = 0; r < 3; r++) jp[r] = rec[3 + 3 * r] * u[0] + rec[4 + 3 * r] * u[1] + rec[5 + 3 * r] * u[2];
        if (kind == GK_ROD) {
#pragma unroll
          for (int r = 0; r < 3; r++)
#pragma unroll
            for (int k = 0; k < 5; k++) jp[r] += GLS(GT_A + (r >= k ? tri(r, k) : tri(k, r))) * GLS(GT_PL + k);
        }
        if (q < GT_LSCAP) {
#pragma unroll
          for (int r = 0; r < 3; r++) GT_LSS(T.b, q, 3 + r) = jp[r];
        } else {
#pragma unroll
          for (int r = 0; r < 3; r++) GRS(base + 25 + r) = jp[r];
        }
      }
      double pMp = 0, pMa = 0, gTp = 0;
#pragma unroll
      for (int k = 0; k < 6; k++) { const double mm = k < 3 ? mt : mr; pMp += mm * T.p[k] * T.p[k]; pMa += mm * T.p[k] * (T.x[k] - (k == 2 ? grav2 : 0.0)); gTp += T.g0[k] * T.p[k]; }
      if (T.lam) {
        double A[15], lam[5], pl[5];
#pragma unroll
        for (int i = 0; i < 15; i++) A[i] = GLS(GT_A + i);
#pragma unroll
        for (int k = 0; k < 5; k++) { lam[k] = GLS(GT_LAM + k); pl[k] = GLS(GT_PL + k); gTp += GLS(GT_HLL + 20 + k) * pl[k]; }
        pMp += gt_symdot5(A, pl, pl); pMa += gt_symdot5(A, pl, lam);
      }
      GT_PUT(T, buf, 0, pMp); GT_PUT(T, buf, 1, pMa); GT_PUT(T, buf, 2, gTp);
    }
    gen_sync();
    PUSH_TOC(6);
    double ls_pMp[NL], ls_pMa[NL], ls_gTp[NL], ls_alpha[NL], ls_lo[NL], ls_hi[NL], ls_best[NL], ls_wprev[NL];
    bool ls_on[NL];
    GT_FOR(li) {
      GTLane& T = t[li];
      ls_pMp[li] = ls_pMa[li] = ls_gTp[li] = 0;
      ls_alpha[li] = 1; ls_lo[li] = 0; ls_hi[li] = -1; ls_best[li] = 1; ls_wprev[li] = 1e300; ls_on[li] = T.active;
      if (!T.active) continue;
#pragma unroll
      for (int c = 0; c < GEN_MAXNB; c++) if ((T.members >> c) & 1) { ls_pMp[li] += GT_GET(buf, c, 0); ls_pMa[li] += GT_GET(buf, c, 1); ls_gTp[li] += GT_GET(buf, c, 2); }
    }
    buf ^= 1;
    bool ls_any = false;
    GT_FOR(li) ls_any = ls_any || ls_on[li];
#pragma clang loop unroll(disable)
    for (int ls = 0; ls < 50 && ls_any; ls++) {
      GT_FOR(li) {
        GTLane& T = t[li];
        if (!ls_on[li]) continue;
        D3IL_STAT(g_stats.ls_iters++);
        PUSH_CNT(3);
        const double alpha = ls_alpha[li];
        double p1 = 0, p2 = 0;
#pragma clang loop unroll(disable)
        for (int q = sub; q < T.cnt; q += NS) {
          const int base = GG_CON + (T.b * GEN_SEG + q) * GREC;
          PUSH_CNT(4);
          double rc[8];      // jar[3] jp[3] Dn fric
          if (q < GT_LSCAP) {
#pragma unroll
            for (int k = 0; k < 8; k++) rc[k] = GT_LSS(T.b, q, k);
          } else {
#pragma unroll
            for (int k = 0; k < 6; k++) rc[k] = GRS(base + 22 + k);
            rc[6] = GRS(base + 19); rc[7] = GRS(base + 20);
          }
          const double jp[3] = {rc[3], rc[4], rc[5]};
          double jt[3] = {rc[0] + alpha * jp[0], rc[1] + alpha * jp[1], rc[2] + alpha * jp[2]}, ft[3], Hc[9];
          const double Dn = rc[6], fric = rc[7];
          gt_cone_eval(jt, Dn, Dn * impr, fric * mu_scale, fric, ft, Hc);
#pragma unroll
          for (int r = 0; r < 3; r++) { p1 -= ft[r] * jp[r];
#pragma unroll
            for (int qq = 0; qq < 3; qq++) p2 += jp[r] * Hc[3 * r + qq] * jp[qq]; }
        }
        if (NS > 1) { double pp[2] = {p1, p2}; gt_pair_sum_n<2>(pp); p1 = pp[0]; p2 = pp[1]; }
        if (T.lam) {
#pragma unroll
          for (int f = 0; f < NFING; f++) {
            const double D = GLS(GT_DF + f);
            double jf = GLS(GT_CA + 3 + f), jpf = 0;
#pragma unroll
            for (int k = 0; k < 5; k++) { const double ak = GLS(GT_A + (3 + f >= k ? tri(3 + f, k) : tri(k, 3 + f))); jf += ak * GLS(GT_LAM + k); jpf += ak * GLS(GT_PL + k); }
            const double jt = jf + alpha * jpf;
            if (D > 0 && jt < 0) { p1 += D * jt * jpf; p2 += D * jpf * jpf; }
          }
        }
        GT_PUT(T, buf, 0, p1); GT_PUT(T, buf, 1, p2);
      }
      gen_sync();
      ls_any = false;
      GT_FOR(li) {
        GTLane& T = t[li];
        if (!ls_on[li]) continue;
        const double alpha = ls_alpha[li], gTp = ls_gTp[li];
        double d1 = ls_pMa[li] + alpha * ls_pMp[li], d2 = ls_pMp[li];
#pragma unroll
        for (int c = 0; c < GEN_MAXNB; c++) if ((T.members >> c) & 1) { d1 += GT_GET(buf, c, 0); d2 += GT_GET(buf, c, 1); }
        ls_best[li] = alpha;
        bool stop = false;
        if (ls == 0 && d1 <= D3IL_TOL.ls_full * fabs(gTp)) stop = true;
        else if (fabs(d1) <= D3IL_TOL.ls_c2 * fabs(gTp) || fabs(d1) <= D3IL_TOL.ls_rel * d2 * alpha || fabs(d1) < 1e-14 * fmax(1.0, fabs(ls_pMa[li]))) stop = true;
        else {
          if (d1 < 0) ls_lo[li] = alpha; else ls_hi[li] = alpha;
          double na = alpha - d1 * rcpd(d2);
          if (ls_hi[li] >= 0) {
            const double wbr = ls_hi[li] - ls_lo[li];
            const bool slow = wbr > 0.5 * ls_wprev[li];
            ls_wprev[li] = wbr;
            if (slow || !(na > ls_lo[li] && na < ls_hi[li])) na = 0.5 * (ls_lo[li] + ls_hi[li]);
          } else if (na <= ls_lo[li]) na = 2 * ls_lo[li] + 1;
          if (na == alpha) stop = true;
          else ls_alpha[li] = na;
        }
        if (stop) ls_on[li] = false;
        ls_any = ls_any || ls_on[li];
      }
      buf ^= 1;
    }
    // ---- step; stopping rule on the island's step size
    GT_FOR(li) {
      GTLane& T = t[li];
      if (!T.active) continue;
      const double best = ls_best[li];
      double smax = 0, xmax = 0;
#pragma unroll
      for (int k = 0; k < 6; k++) { const double dxk = best * T.p[k]; T.x[k] += dxk; smax = fmax(smax, fabs(dxk)); xmax = fmax(xmax, fabs(T.x[k])); }
#pragma unroll
      for (int k = 0; k < 6; k++) GLS(GL_X + 6 * T.b + k) = T.x[k];
      if (T.lam) {
#pragma unroll
        for (int k = 0; k < 5; k++) { const double dl = best * GLS(GT_PL + k), ln = GLS(GT_LAM + k) + dl; GLS(GT_LAM + k) = ln; smax = fmax(smax, fabs(dl)); xmax = fmax(xmax, fabs(ln)); }
      }
      GT_PUT(T, buf, 0, smax); GT_PUT(T, buf, 1, xmax); GT_PUT(T, buf, 2, T.failed ? 1.0 : 0.0);
    }
    gen_sync();
    any = false;
    GT_FOR(li) {
      GTLane& T = t[li];
      if (!T.active) continue;
      double smax = 0, xmax = 0, bad = 0;
#pragma unroll
      for (int c = 0; c < GEN_MAXNB; c++) if ((T.members >> c) & 1) { smax = fmax(smax, GT_GET(buf, c, 0)); xmax = fmax(xmax, GT_GET(buf, c, 1)); bad = fmax(bad, GT_GET(buf, c, 2)); }
      if (bad != 0) { T.failed = true; T.active = false; }
      else if (smax <= 1e-12 * (1 + xmax) || (ls_best[li] == 1.0 && smax <= D3IL_TOL.step_rel * (1 + xmax))) T.active = false;
      any = any || T.active;
    }
    buf ^= 1;
    PUSH_TOC(7);
  }
  GT_FOR(li) {
    GTLane& T = t[li];
    if (!T.fast) continue;
    if (T.active || T.failed) fl |= F_SOLVER_FAIL;
    if (T.cnt > 0 || T.members != (1u << T.b)) {      // the lane took part in a solve: its iterate goes back into the body axes of the free joint
      gt_to_body(sc, T.b, T.x);
#pragma unroll
      for (int k = 3; k < 6; k++) GLS(GL_X + 6 * T.b + k) = T.x[k];
    }
    if (T.lam) {
#pragma unroll
      for (int k = 0; k < 5; k++) GLS(GL_TR + 1 + k) = GLS(GT_LAM + k);
      GLS(GL_TR) = 2.0;
    }
  }
  gen_sync();
  return fl;
}


// ---- a cube on its own (an environment without cube <-> cube and rod contacts: every cube rests on static boxes only - the resting regime and most sub-steps
// of an evaluation run): the single-node case of the tree solver as a function of its own.  No tree analysis, no messages, no exchange slots - and an
// order fewer registers than gen_tree_solve, so it is built INTO the step kernel: an environment that needs nothing else never calls the big solver (whose
// callee-saved register block is most of the engine's scratch traffic, profiles/r05/README.md).  Same cost function, point form, line search and stopping
// rules; the aref / D of a record are kept in registers' reach (LDS line-search slots) and in the record.  NS, sub as in gen_tree_solve.
template <int NS>
D3IL_HD unsigned gen_lone_solve(const GenConsts& gc_, const PushScratch sc, int b, bool warm_valid, int sub) {
  D3IL_GEN_CONSTS(gc_, gc);
  const int cnt = (int)((unsigned)GLS(GL_INFO + b) & 31u);
  double x[6];
  if (cnt == 0) {                                   // in free flight: x = a0
#pragma unroll
    for (int k = 0; k < 6; k++) GLS(GL_X + 6 * b + k) = GLS(GL_A0 + 6 * b + k);
    return 0u;
  }
  const double impr = gc.impratio, mu_scale = sqrt(1 / fmax(1e-15, impr)), mt = gc.box_mass, mr = gc.box_inertia;
  const double grav2 = GLS(GL_A0 + 2);
#pragma unroll
  for (int k = 0; k < 6; k++) x[k] = warm_valid ? GWARM(6 * b + k) : GLS(GL_A0 + 6 * b + k);
  gt_to_world(sc, b, x);
  double pc[3];
#pragma unroll
  for (int k = 0; k < 3; k++) pc[k] = GLS(GL_POS + 3 * b + k);
  {   // reference acceleration and regularisation of the records (all of them cube <-> static box)
    double vel[6];
#pragma unroll
    for (int k = 0; k < 6; k++) vel[k] = GLS(GL_VEL + 6 * b + k);
    gt_to_world(sc, b, vel);
#pragma clang loop unroll(disable)
    for (int q = sub; q < cnt; q += NS) {
      const int base = GG_CON + (b * GEN_SEG + q) * GREC;
      double rec[16];
#pragma unroll
      for (int k = 0; k < 16; k++) rec[k] = GRS(base + k);
      const int set = (int)rec[14];
      const double sg = GRS(base + 21);
      const double r1[3] = {rec[0] - pc[0], rec[1] - pc[1], rec[2] - pc[2]};
      double u[3] = {0, 0, 0}, v[3];
      gt_point(vel, r1, sg, u);
#pragma unroll
      for (int r = 0; r < 3; r++) v[r] = rec[3 + 3 * r] * u[0] + rec[4 + 3 * r] * u[1] + rec[5 + 3 * r] * u[2];
      const double dist = rec[12], imp = impedance(gc.ct_solimp[set], dist);
      const double Dn = 1 / fmax(1e-15, (1 - imp) / imp * gc.box_invw_t);
      GRS(base + 16) = -gc.ct_B[set] * v[0] - gc.ct_K[set] * imp * dist;
      GRS(base + 17) = -gc.ct_B[set] * v[1]; GRS(base + 18) = -gc.ct_B[set] * v[2];
      GRS(base + 19) = Dn;
      if (q < GT_LSCAP) { GT_LSS(b, q, 6) = Dn; GT_LSS(b, q, 7) = GRS(base + 20); }
    }
    D3IL_STAT(g_stats.newton_calls++);
  }
  bool converged = false, failed = false;
#pragma clang loop unroll(disable)
  for (int it = 0; it < 60 && !converged; it++) {
    D3IL_STAT(g_stats.newton_iters++);
    double g[6], H[21];
#pragma unroll
    for (int i = 0; i < 21; i++) H[i] = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) { const double mm = sub == 0 ? (k < 3 ? mt : mr) : 0.0; g[k] = mm * (x[k] - (k == 2 ? grav2 : 0.0)); H[tri(k, k)] = mm; }
#pragma clang loop unroll(disable)
    for (int q = sub; q < cnt; q += NS) {
      const int base = GG_CON + (b * GEN_SEG + q) * GREC;
      double rec[22];
#pragma unroll
      for (int k = 0; k < 22; k++) rec[k] = GRS(base + k);
      const double sg = rec[21];
      const double r1[3] = {rec[0] - pc[0], rec[1] - pc[1], rec[2] - pc[2]};
      double u[3] = {0, 0, 0}, jar[3], force[3], Hc[9];
      gt_point(x, r1, sg, u);
#pragma unroll
      for (int r = 0; r < 3; r++) jar[r] = rec[3 + 3 * r] * u[0] + rec[4 + 3 * r] * u[1] + rec[5 + 3 * r] * u[2] - rec[16 + r];
      if (q < GT_LSCAP) {
#pragma unroll
        for (int r = 0; r < 3; r++) GT_LSS(b, q, r) = jar[r];
      } else {
#pragma unroll
        for (int r = 0; r < 3; r++) GRS(base + 22 + r) = jar[r];
      }
      const double Dn = rec[19], fric = rec[20];
      gt_cone_eval(jar, Dn, Dn * impr, fric * mu_scale, fric, force, Hc);
      if (force[0] == 0 && force[1] == 0 && force[2] == 0) continue;
      double w[3], S[9], P[9];
#pragma unroll
      for (int k = 0; k < 3; k++) w[k] = sg * (rec[3 + k] * force[0] + rec[6 + k] * force[1] + rec[9 + k] * force[2]);
      g[0] -= w[0]; g[1] -= w[1]; g[2] -= w[2];
      g[3] -= r1[1] * w[2] - r1[2] * w[1]; g[4] -= r1[2] * w[0] - r1[0] * w[2]; g[5] -= r1[0] * w[1] - r1[1] * w[0];
      gt_world_S(rec + 3, Hc, S);
      gt_SxR(S, r1, P);
      gt_add_diag(H, S, P, r1);
    }
    if (NS > 1) { gt_pair_sum_n<6>(g); gt_pair_sum_n<21>(H); }
    {
      double gm = 0;
#pragma unroll
      for (int k = 0; k < 6; k++) gm = fmax(gm, fabs(g[k]));
      if (gm <= PUSH_GRAD_TOL) { converged = true; break; }
    }
    double dd[6], idd[6], p[6];
    if (!gt_ldl_n<6>(H, dd, idd)) failed = true;
#pragma unroll
    for (int k = 0; k < 6; k++) p[k] = -g[k];
    gt_ldl_solve_n<6>(H, idd, p);
    double pMp = 0, pMa = 0, gTp = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) { const double mm = k < 3 ? mt : mr; pMp += mm * p[k] * p[k]; pMa += mm * p[k] * (x[k] - (k == 2 ? grav2 : 0.0)); gTp += g[k] * p[k]; }
#pragma clang loop unroll(disable)
    for (int q = sub; q < cnt; q += NS) {      // directional derivatives of the rows
      const int base = GG_CON + (b * GEN_SEG + q) * GREC;
      double rec[12];
#pragma unroll
      for (int k = 0; k < 12; k++) rec[k] = GRS(base + k);
      const double sg = GRS(base + 21);
      const double r1[3] = {rec[0] - pc[0], rec[1] - pc[1], rec[2] - pc[2]};
      double u[3] = {0, 0, 0}, jp[3];
      gt_point(p, r1, sg, u);
#pragma unroll
      for (int r = 0; r < 3; r++) jp[r] = rec[3 + 3 * r] * u[0] + rec[4 + 3 * r] * u[1] + rec[5 + 3 * r] * u[2];
      if (q < GT_LSCAP) {
#pragma unroll
        for (int r = 0; r < 3; r++) GT_LSS(b, q, 3 + r) = jp[r];
      } else {
#pragma unroll
        for (int r = 0; r < 3; r++) GRS(base + 25 + r) = jp[r];
      }
    }
    double alpha = 1, lo = 0, hi = -1, best = 1, wprev = 1e300;
#pragma clang loop unroll(disable)
    for (int ls = 0; ls < 50; ls++) {
      D3IL_STAT(g_stats.ls_iters++);
      double p1 = 0, p2 = 0;
#pragma clang loop unroll(disable)
      for (int q = sub; q < cnt; q += NS) {
        const int base = GG_CON + (b * GEN_SEG + q) * GREC;
        double rc[8];      // jar[3] jp[3] Dn fric
        if (q < GT_LSCAP) {
#pragma unroll
          for (int k = 0; k < 8; k++) rc[k] = GT_LSS(b, q, k);
        } else {
#pragma unroll
          for (int k = 0; k < 6; k++) rc[k] = GRS(base + 22 + k);
          rc[6] = GRS(base + 19); rc[7] = GRS(base + 20);
        }
        const double jp[3] = {rc[3], rc[4], rc[5]};
        double jt[3] = {rc[0] + alpha * jp[0], rc[1] + alpha * jp[1], rc[2] + alpha * jp[2]}, ft[3], Hc[9];
        gt_cone_eval(jt, rc[6], rc[6] * impr, rc[7] * mu_scale, rc[7], ft, Hc);
#pragma unroll
        for (int r = 0; r < 3; r++) { p1 -= ft[r] * jp[r];
#pragma unroll
          for (int qq = 0; qq < 3; qq++) p2 += jp[r] * Hc[3 * r + qq] * jp[qq]; }
      }
      if (NS > 1) { double pp[2] = {p1, p2}; gt_pair_sum_n<2>(pp); p1 = pp[0]; p2 = pp[1]; }
      const double d1 = pMa + alpha * pMp + p1, d2 = pMp + p2;
      best = alpha;
      if (ls == 0 && d1 <= D3IL_TOL.ls_full * fabs(gTp)) break;
      if (fabs(d1) <= D3IL_TOL.ls_c2 * fabs(gTp) || fabs(d1) <= D3IL_TOL.ls_rel * d2 * alpha || fabs(d1) < 1e-14 * fmax(1.0, fabs(pMa))) break;
      if (d1 < 0) lo = alpha; else hi = alpha;
      double na = alpha - d1 * rcpd(d2);
      if (hi >= 0) {
        const double wbr = hi - lo;
        const bool slow = wbr > 0.5 * wprev;
        wprev = wbr;
        if (slow || !(na > lo && na < hi)) na = 0.5 * (lo + hi);
      } else if (na <= lo) na = 2 * lo + 1;
      if (na == alpha) break;
      alpha = na;
    }
    double smax = 0, xmax = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) { const double dxk = best * p[k]; x[k] += dxk; smax = fmax(smax, fabs(dxk)); xmax = fmax(xmax, fabs(x[k])); }
    if (failed) break;
    if (smax <= 1e-12 * (1 + xmax) || (best == 1.0 && smax <= D3IL_TOL.step_rel * (1 + xmax))) converged = true;
  }
  gt_to_body(sc, b, x);
#pragma unroll
  for (int k = 0; k < 6; k++) GLS(GL_X + 6 * b + k) = x[k];
  return (converged && !failed) ? 0u : (unsigned)F_SOLVER_FAIL;
}

}  // namespace d3il
#if defined(__clang__)
#pragma clang fp contract(fast)      // (what hipcc compiles the rest of the translation unit with)
#endif
