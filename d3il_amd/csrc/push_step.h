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

namespace d3il {

constexpr int PUSH_NB = 2;              // free cubes
constexpr int PUSH_NV = 21;             // solver dof order: cube1[6] cube2[6] arm[9]
constexpr int PUSH_ARM0 = 12;
constexpr int PUSH_MAXCON = 24;
constexpr int PUSH_NH = PUSH_NV * (PUSH_NV + 1) / 2;   // 231
constexpr int PUSH_MAXIT = 40;

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

struct BoxState { double pos[3], quat[4], vel[6]; };
struct PushState {
  EnvState arm;
  BoxState box[PUSH_NB];
  double warm[PUSH_NV];
};

// per-lane views of the two scratch areas: h = Hessian (LDS on the device), g = everything else (HBM)
struct PushScratch {
  double* h; int hs;
  double* g; int gs;
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
constexpr int PG_SIZE = PG_GRAD + PUSH_NV;              // 774

#define PGS(i) sc.g[(long)(i) * sc.gs]
#define PHS(i) sc.h[(long)(i) * sc.hs]

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
D3IL_HD int box_box(const double* p1, const double* R1, const double* s1, const double* p2, const double* R2, const double* s2,
                    double margin, double (*out)[7], int cap) {
  const double FUDGE = 1.05;
  double A[3][3], B[3][3], d[3], Cm[3][3], Q[3][3], dA[3], dB[3];
  for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) { A[i][k] = R1[3 * k + i]; B[i][k] = R2[3 * k + i]; }
  for (int k = 0; k < 3; k++) d[k] = p2[k] - p1[k];
  for (int i = 0; i < 3; i++) {
    dA[i] = dot3(d, A[i]); dB[i] = dot3(d, B[i]);
    for (int j = 0; j < 3; j++) { Cm[i][j] = dot3(A[i], B[j]); Q[i][j] = fabs(Cm[i][j]); }
  }
  double best = -1e300; int code = -1; double nsign = 1;
  for (int i = 0; i < 3; i++) {
    double sep = fabs(dA[i]) - (s1[i] + s2[0] * Q[i][0] + s2[1] * Q[i][1] + s2[2] * Q[i][2]);
    if (sep > margin) return 0;
    if (sep > best + 1e-10) { best = sep; code = i; nsign = dA[i] < 0 ? -1 : 1; }
  }
  for (int j = 0; j < 3; j++) {
    double sep = fabs(dB[j]) - (s2[j] + s1[0] * Q[0][j] + s1[1] * Q[1][j] + s1[2] * Q[2][j]);
    if (sep > margin) return 0;
    if (sep > best + 1e-10) { best = sep; code = 3 + j; nsign = dB[j] < 0 ? -1 : 1; }
  }
  double en[3] = {0, 0, 0};
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
    int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    double l2 = 1 - Cm[i][j] * Cm[i][j];
    if (l2 < 1e-10) continue;
    double l = sqrt(l2);
    double proj = dA[i2] * Cm[i1][j] - dA[i1] * Cm[i2][j];
    double ra = s1[i1] * Q[i2][j] + s1[i2] * Q[i1][j], rb = s2[j1] * Q[i][j2] + s2[j2] * Q[i][j1];
    double sep = (fabs(proj) - (ra + rb)) / l;
    if (sep > margin) return 0;
    if (sep * FUDGE > best + 1e-10 && sep > best) {
      best = sep; code = 6 + 3 * i + j;
      double Lx[3]; cross3(A[i], B[j], Lx);
      double sg = proj < 0 ? -1 : 1;
      for (int k = 0; k < 3; k++) en[k] = sg * Lx[k] / l;
    }
  }
  if (code >= 6) {
    int i = (code - 6) / 3, j = (code - 6) % 3;
    double pa[3], pb[3];
    for (int k = 0; k < 3; k++) { pa[k] = p1[k]; pb[k] = p2[k]; }
    for (int a = 0; a < 3; a++) if (a != i) { double sg = dot3(en, A[a]) > 0 ? 1 : -1; for (int k = 0; k < 3; k++) pa[k] += sg * s1[a] * A[a][k]; }
    for (int b = 0; b < 3; b++) if (b != j) { double sg = dot3(en, B[b]) > 0 ? -1 : 1; for (int k = 0; k < 3; k++) pb[k] += sg * s2[b] * B[b][k]; }
    double w[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
    double cc = Cm[i][j], wa = dot3(w, A[i]), wb = dot3(w, B[j]), den = 1 - cc * cc;
    double al = (wa - cc * wb) / den, be = (cc * wa - wb) / den;
    if (cap < 1) return 0;
    out[0][0] = best;
    for (int k = 0; k < 3; k++) { out[0][1 + k] = 0.5 * (pa[k] + al * A[i][k] + pb[k] + be * B[j][k]); out[0][4 + k] = en[k]; }
    return 1;
  }
  bool ref2 = code >= 3; int ax = ref2 ? code - 3 : code;
  const double* pr = ref2 ? p2 : p1; const double* pi = ref2 ? p1 : p2;
  const double* sr = ref2 ? s2 : s1; const double* si = ref2 ? s1 : s2;
  double (*Ar)[3] = ref2 ? B : A; double (*Ai)[3] = ref2 ? A : B;
  double n[3], sgn = ref2 ? -nsign : nsign;
  for (int k = 0; k < 3; k++) n[k] = sgn * Ar[ax][k];
  int kin = 0; double bestdot = -1;
  for (int k = 0; k < 3; k++) { double t = fabs(dot3(n, Ai[k])); if (t > bestdot) { bestdot = t; kin = k; } }
  double sgi = dot3(n, Ai[kin]) > 0 ? -1 : 1;
  int k1 = (kin + 1) % 3, k2 = (kin + 2) % 3, a1 = (ax + 1) % 3, a2 = (ax + 2) % 3;
  double poly[16][3], tmp[16][3]; int np = 4;
  for (int v = 0; v < 4; v++) {
    double c0 = (v == 0 || v == 3) ? 1.0 : -1.0, c1 = v < 2 ? 1.0 : -1.0, x[3];
    for (int k = 0; k < 3; k++) x[k] = pi[k] + sgi * si[kin] * Ai[kin][k] + c0 * si[k1] * Ai[k1][k] + c1 * si[k2] * Ai[k2][k] - pr[k];
    poly[v][0] = dot3(x, Ar[a1]); poly[v][1] = dot3(x, Ar[a2]); poly[v][2] = dot3(x, n) - sr[ax];
  }
  for (int side = 0; side < 4; side++) {
    int cdim = side >> 1; double sg = (side & 1) ? -1 : 1, lim = sr[cdim ? a2 : a1];
    int nn = 0;
    for (int v = 0; v < np; v++) {
      int vn = v + 1 == np ? 0 : v + 1;
      double fp = sg * poly[v][cdim] - lim, fq = sg * poly[vn][cdim] - lim;
      if (fp <= 0) { for (int k = 0; k < 3; k++) tmp[nn][k] = poly[v][k]; nn++; }
      if ((fp <= 0) != (fq <= 0)) { double t = fp / (fp - fq); for (int k = 0; k < 3; k++) tmp[nn][k] = poly[v][k] + t * (poly[vn][k] - poly[v][k]); nn++; }
    }
    np = nn;
    for (int v = 0; v < np; v++) for (int k = 0; k < 3; k++) poly[v][k] = tmp[v][k];
    if (np == 0) return 0;
  }
  int cnt = 0;
  for (int v = 0; v < np && cnt < cap; v++) {
    double w = poly[v][2];
    if (w >= margin) continue;
    bool dup = false;
    for (int q = 0; q < v; q++) if (fabs(poly[q][0] - poly[v][0]) + fabs(poly[q][1] - poly[v][1]) < 1e-12 && poly[q][2] < margin) dup = true;
    if (dup) continue;
    out[cnt][0] = w;
    for (int k = 0; k < 3; k++) {
      out[cnt][1 + k] = pr[k] + poly[v][0] * Ar[a1][k] + poly[v][1] * Ar[a2][k] + (sr[ax] + 0.5 * w) * n[k];
      out[cnt][4 + k] = ref2 ? -n[k] : n[k];
    }
    cnt++;
  }
  return cnt;
}

// rod (cylinder, axis u through pc, radius rad, half length half) against a box: closest points of the axis segment
// and the box, minus the radius (side contacts; the flat end caps are not modelled).  Normal from the box to the rod.
D3IL_HD bool cyl_box(const double* pc, const double* axis, double rad, double half, const double* pb, const double* Rb, const double* sb,
                     double margin, double* out) {
  double c[3], u[3], rel[3] = {pc[0] - pb[0], pc[1] - pb[1], pc[2] - pb[2]};
  for (int i = 0; i < 3; i++) { double col[3] = {Rb[i], Rb[3 + i], Rb[6 + i]}; c[i] = dot3(rel, col); u[i] = dot3(axis, col); }
  double T[8]; int nt = 0;
  T[nt++] = -half; T[nt++] = half;
  for (int i = 0; i < 3; i++) if (fabs(u[i]) > 1e-14) for (int sg = -1; sg <= 1; sg += 2) { double t = (sg * sb[i] - c[i]) / u[i]; if (t > -half && t < half) T[nt++] = t; }
  for (int a = 1; a < nt; a++) { double v = T[a]; int b = a - 1; while (b >= 0 && T[b] > v) { T[b + 1] = T[b]; b--; } T[b + 1] = v; }
  double G[8];
  for (int k = 0; k < nt; k++) {
    double g = 0;
    for (int i = 0; i < 3; i++) { double x = c[i] + T[k] * u[i], cl = x > sb[i] ? sb[i] : (x < -sb[i] ? -sb[i] : x); g += u[i] * (x - cl); }
    G[k] = g;
  }
  const double tol = 1e-13;
  double tm, tp;
  { int k = 0; while (k < nt && G[k] < -tol) k++;
    if (k == 0) tm = T[0]; else if (k == nt) tm = T[nt - 1];
    else tm = G[k] > tol ? T[k - 1] + (T[k] - T[k - 1]) * (-G[k - 1]) / (G[k] - G[k - 1]) : T[k]; }
  { int k = nt - 1; while (k >= 0 && G[k] > tol) k--;
    if (k == nt - 1) tp = T[nt - 1]; else if (k < 0) tp = T[0];
    else tp = G[k] < -tol ? T[k] + (T[k + 1] - T[k]) * (-G[k]) / (G[k + 1] - G[k]) : T[k]; }
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
// Jacobian row of a cube (6 entries: linear world, angular body axes) for a world direction f at the world point p
D3IL_HD void box_row(const double* R, const double* bpos, const double* p, const double* f, double* row) {
  double r[3] = {p[0] - bpos[0], p[1] - bpos[1], p[2] - bpos[2]}, rxf[3];
  cross3(r, f, rxf);
  row[0] = f[0]; row[1] = f[1]; row[2] = f[2];
  row[3] = R[0] * rxf[0] + R[3] * rxf[1] + R[6] * rxf[2];
  row[4] = R[1] * rxf[0] + R[4] * rxf[1] + R[7] * rxf[2];
  row[5] = R[2] * rxf[0] + R[5] * rxf[1] + R[8] * rxf[2];
}

// Sparse row of one contact direction over the 21 solver dofs: up to two blocks (offset, length, values).
struct SRow { int o1, n1, o2, n2; double v1[7], v2[7]; };

// builds the three rows (normal, t1, t2) of contact `ci` from its record; the relative velocity is body2 - body1 with the
// normal pointing from geom1 to geom2: slab(1) -> cube(2); cube1(1) -> cube2(2); cube(1) -> rod(2)
D3IL_HD void contact_rows(const PushScratch& sc, int ci, const double (*Rb)[9], const BoxState* box, SRow* rows) {
  int base = PG_CON + ci * PREC;
  double p[3] = {PGS(base), PGS(base + 1), PGS(base + 2)};
  int kind = (int)PGS(base + 13), cube = (int)PGS(base + 14);
  for (int r = 0; r < 3; r++) {
    double f[3] = {PGS(base + 3 + 3 * r), PGS(base + 4 + 3 * r), PGS(base + 5 + 3 * r)};
    SRow& s = rows[r];
    if (kind == CK_SLAB) {
      s.o1 = 6 * cube; s.n1 = 6; s.n2 = 0; s.o2 = 0;
      box_row(Rb[cube], box[cube].pos, p, f, s.v1);
    } else if (kind == CK_BOXBOX) {
      s.o1 = 0; s.n1 = 6; s.o2 = 6; s.n2 = 6;
      box_row(Rb[0], box[0].pos, p, f, s.v1);
      for (int k = 0; k < 6; k++) s.v1[k] = -s.v1[k];
      box_row(Rb[1], box[1].pos, p, f, s.v2);
    } else {
      s.o1 = 6 * cube; s.n1 = 6; s.o2 = PUSH_ARM0; s.n2 = 7;
      box_row(Rb[cube], box[cube].pos, p, f, s.v1);
      for (int k = 0; k < 6; k++) s.v1[k] = -s.v1[k];
      for (int k = 0; k < 7; k++) s.v2[k] = PGS(PG_JA + cube * 21 + r * 7 + k);
    }
  }
}
D3IL_HD double srow_dot(const SRow& s, const double* x) {
  double a = 0;
  for (int k = 0; k < s.n1; k++) a += s.v1[k] * x[s.o1 + k];
  for (int k = 0; k < s.n2; k++) a += s.v2[k] * x[s.o2 + k];
  return a;
}

// elliptic cone (condim 3, friction mu_geom on both tangents): force and Hessian block at row residuals jar.
// Returns the cost.  zone: 0 top (free), 1 bottom (quadratic), 2 middle.
D3IL_HD double cone_eval(const double* jar, double Dn, double Dt, double mu, double fric, double* force, double* Hc /* 3x3 or null */) {
  double U0 = jar[0] * mu, U1 = jar[1] * fric, U2 = jar[2] * fric;
  double N = U0, T = sqrt(U1 * U1 + U2 * U2);
  if (Hc) for (int i = 0; i < 9; i++) Hc[i] = 0;
  if (N >= mu * T || (T <= 0 && N >= 0)) { force[0] = force[1] = force[2] = 0; return 0; }
  if (mu * N + T <= 0 || (T <= 0 && N < 0)) {
    force[0] = -Dn * jar[0]; force[1] = -Dt * jar[1]; force[2] = -Dt * jar[2];
    if (Hc) { Hc[0] = Dn; Hc[4] = Dt; Hc[8] = Dt; }
    return 0.5 * (Dn * jar[0] * jar[0] + Dt * jar[1] * jar[1] + Dt * jar[2] * jar[2]);
  }
  double Dm = Dn / fmax(1e-15, mu * mu * (1 + mu * mu)), NmT = N - mu * T;
  double g[3] = {mu, -mu * fric * U1 / T, -mu * fric * U2 / T};
  for (int j = 0; j < 3; j++) force[j] = -Dm * NmT * g[j];
  if (Hc) {
    double U[3] = {0, U1, U2}, iT = 1 / T, iT3 = iT * iT * iT;
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) {
      double h = g[a] * g[b];
      if (a > 0 && b > 0) h += NmT * (-mu) * fric * fric * ((a == b ? iT : 0) - U[a] * U[b] * iT3);
      Hc[3 * a + b] = Dm * h;
    }
  }
  return 0.5 * Dm * NmT * NmT;
}

// ------------------------------------------------------------------------------------------------ skyline Cholesky in the h area
// first[i] = first column of row i inside the envelope (wave-uniform): cube1 rows 0, cube2 rows (bb ? 0 : 6),
// arm rows (rod on cube1 ? 0 : rod on cube2 ? 6 : 12)
D3IL_HD int sky_first(int i, bool bb, bool rod1, bool rod2) {
  if (i < 6) return 0;
  if (i < 12) return bb ? 0 : 6;
  return rod1 ? 0 : (rod2 ? 6 : 12);
}
D3IL_HD bool sky_chol(const PushScratch& sc, bool bb, bool rod1, bool rod2) {
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
D3IL_HD void sky_solve(const PushScratch& sc, bool bb, bool rod1, bool rod2, double* x) {
  for (int i = 0; i < PUSH_NV; i++) {
    int fi = sky_first(i, bb, rod1, rod2);
    double s = x[i];
    for (int k = fi; k < i; k++) s -= PHS(tri(i, k)) * x[k];
    x[i] = s / PHS(tri(i, i));
  }
  for (int i = PUSH_NV - 1; i >= 0; i--) {
    double xi = x[i] / PHS(tri(i, i));
    x[i] = xi;
    int fi = sky_first(i, bb, rod1, rod2);
    for (int k = fi; k < i; k++) x[k] -= PHS(tri(i, k)) * xi;
  }
}

// wave-level OR of a per-lane predicate (host: identity)
D3IL_HD bool wave_any(bool p) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __any(p) != 0;
#else
  return p;
#endif
}

// ------------------------------------------------------------------------------------------------ the physics sub-step
struct LimitRow { double sign, D, aref; };

template <class C>
D3IL_HD void push_physics_substep(const C& c0, const PushConsts& pc, PushState& ps, const PushScratch& sc, const double* tau, const double* ffing) {
  EnvState& st = ps.arm;
  double fs[NDOF], tcp_new[3], rodc[3], rodu[3];
  double h = 0;
  // ---- arm: dynamics, smooth force, read-backs (as panda_step.h physics_substep)
  {
    DynOut dyn;
    dynamics(c0, st.q, st.v, dyn);
    D3IL_REFRESH(c0, c);
    h = c.timestep;
#pragma unroll
    for (int k = 0; k < NARM; k++) fs[k] = clampd(tau[k] + st.bias[k], c.force_lo[k], c.force_hi[k]) - dyn.bias[k];
#pragma unroll
    for (int k = 0; k < NFING; k++) fs[NARM + k] = clampd(ffing[k], c.force_lo[NARM + k], c.force_hi[NARM + k]) - dyn.bias[NARM + k] - c.f_damping[k] * st.v[NARM + k];
#pragma unroll
    for (int k = 0; k < NARM; k++) st.bias[k] = dyn.bias[k];
    double t[3]; mulE(dyn.R7, c.tcp7, t);
    tcp_new[0] = dyn.p7[0] + t[0]; tcp_new[1] = dyn.p7[1] + t[1]; tcp_new[2] = dyn.p7[2] + t[2];
    mulE(dyn.R7, c.rod_c7, rodc); rodc[0] += dyn.p7[0]; rodc[1] += dyn.p7[1]; rodc[2] += dyn.p7[2];
    mulE(dyn.R7, c.rod_u7, rodu);
    for (int i = 0; i < 45; i++) PGS(PG_M + i) = dyn.M[i];
    // qacc_smooth of the arm
    double L[45], d[NDOF], id[NDOF], a0[NDOF];
    if (!ldl9(dyn.M, L, d, id)) st.flags |= F_SOLVER_FAIL;
#pragma unroll
    for (int k = 0; k < NDOF; k++) a0[k] = fs[k];
    ldl9_solve(L, id, a0);
    for (int k = 0; k < NDOF; k++) PGS(PG_A0 + PUSH_ARM0 + k) = a0[k];
  }
  st.tcp[0] = tcp_new[0]; st.tcp[1] = tcp_new[1]; st.tcp[2] = tcp_new[2];
  D3IL_REFRESH(c0, c);
  // ---- cubes: kinematics and smooth acceleration (gravity; isotropic inertia has no gyroscopic term)
  double Rb[PUSH_NB][9];
  for (int b = 0; b < PUSH_NB; b++) {
    quat2mat(ps.box[b].quat, Rb[b]);
    for (int k = 0; k < 6; k++) PGS(PG_A0 + 6 * b + k) = k < 3 ? c.gravity[k] : 0.0;
    if (fabs(ps.box[b].pos[0] - pc.slab_c[0][0]) > pc.slab_h[0][0] - 0.06 || fabs(ps.box[b].pos[1] - pc.slab_c[0][1]) > pc.slab_h[0][1] - 0.06) st.flags |= PF_OFF_TABLE;
  }
  // ---- collision -> contact records
  int ncon = 0; bool has_bb = false, has_rod[2] = {false, false};
  {
    const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    double rec[8][7];
    for (int b = 0; b < PUSH_NB; b++) for (int s = 0; s < 2; s++) {
      int n = box_box(pc.slab_c[s], I3, pc.slab_h[s], ps.box[b].pos, Rb[b], pc.box_half, 0.0, rec, 8);
      for (int i = 0; i < n; i++) {
        if (ncon >= PUSH_MAXCON) { st.flags |= PF_CON_OVERFLOW; break; }
        int base = PG_CON + ncon * PREC;
        for (int k = 0; k < 3; k++) { PGS(base + k) = rec[i][1 + k]; PGS(base + 3 + k) = rec[i][4 + k]; }
        PGS(base + 12) = rec[i][0]; PGS(base + 13) = CK_SLAB; PGS(base + 14) = b;
        ncon++;
      }
    }
    {
      int n = box_box(ps.box[0].pos, Rb[0], pc.box_half, ps.box[1].pos, Rb[1], pc.box_half, 0.0, rec, 8);
      for (int i = 0; i < n; i++) {
        if (ncon >= PUSH_MAXCON) { st.flags |= PF_CON_OVERFLOW; break; }
        int base = PG_CON + ncon * PREC;
        for (int k = 0; k < 3; k++) { PGS(base + k) = rec[i][1 + k]; PGS(base + 3 + k) = rec[i][4 + k]; }
        PGS(base + 12) = rec[i][0]; PGS(base + 13) = CK_BOXBOX; PGS(base + 14) = 0;
        ncon++; has_bb = true;
      }
    }
    for (int b = 0; b < PUSH_NB; b++) {
      double r1[7];
      if (cyl_box(rodc, rodu, c.rod_r, c.rod_h, ps.box[b].pos, Rb[b], pc.box_half, 0.0, r1)) {
        if (ncon >= PUSH_MAXCON) { st.flags |= PF_CON_OVERFLOW; continue; }
        int base = PG_CON + ncon * PREC;
        for (int k = 0; k < 3; k++) { PGS(base + k) = r1[1 + k]; PGS(base + 3 + k) = r1[4 + k]; }
        PGS(base + 12) = r1[0]; PGS(base + 13) = CK_ROD; PGS(base + 14) = b;
        ncon++; has_rod[b] = true;
      }
    }
  }
  // arm Jacobian rows of the rod contacts (world joint axes / origins from the arm chain)
  if (has_rod[0] || has_rod[1]) {
    double sn[NARM], cs[NARM], R7[9], p7[3], ax[NARM][3], og[NARM][3];
    for (int k = 0; k < NARM; k++) { sn[k] = sin(st.q[k]); cs[k] = cos(st.q[k]); }
    world_chain(c0, sn, cs, R7, p7, ax, og);
    for (int ci = 0; ci < ncon; ci++) {
      int base = PG_CON + ci * PREC;
      if ((int)PGS(base + 13) != CK_ROD) continue;
      int b = (int)PGS(base + 14);
      double p[3] = {PGS(base), PGS(base + 1), PGS(base + 2)}, n[3] = {PGS(base + 3), PGS(base + 4), PGS(base + 5)}, t1[3], t2[3];
      make_frame(n, t1, t2);
      for (int k = 0; k < NARM; k++) {
        double dd[3] = {p[0] - og[k][0], p[1] - og[k][1], p[2] - og[k][2]}, col[3];
        cross3(ax[k], dd, col);
        PGS(PG_JA + b * 21 + k) = dot3(n, col); PGS(PG_JA + b * 21 + 7 + k) = dot3(t1, col); PGS(PG_JA + b * 21 + 14 + k) = dot3(t2, col);
      }
    }
  }
  // ---- per-contact frame, reference acceleration and regularisation
  double vel21[PUSH_NV];
  for (int b = 0; b < PUSH_NB; b++) for (int k = 0; k < 6; k++) vel21[6 * b + k] = ps.box[b].vel[k];
  for (int k = 0; k < NDOF; k++) vel21[PUSH_ARM0 + k] = st.v[k];
  for (int ci = 0; ci < ncon; ci++) {
    int base = PG_CON + ci * PREC;
    double n[3] = {PGS(base + 3), PGS(base + 4), PGS(base + 5)}, t1[3], t2[3];
    make_frame(n, t1, t2);
    for (int k = 0; k < 3; k++) { PGS(base + 6 + k) = t1[k]; PGS(base + 9 + k) = t2[k]; }
    int kind = (int)PGS(base + 13), set = kind == CK_SLAB ? 0 : 1;
    double dist = PGS(base + 12);
    double imp = impedance(pc.ct_solimp[set], dist);
    double invw = kind == CK_SLAB ? pc.box_invw_t : (kind == CK_BOXBOX ? 2 * pc.box_invw_t : pc.box_invw_t + c.rod_invweight0);
    double Rn = fmax(1e-15, (1 - imp) / imp * invw);
    SRow rows[3];
    contact_rows(sc, ci, Rb, ps.box, rows);
    double v0 = srow_dot(rows[0], vel21), v1 = srow_dot(rows[1], vel21), v2 = srow_dot(rows[2], vel21);
    PGS(base + 15) = -pc.ct_B[set] * v0 - pc.ct_K[set] * imp * dist;
    PGS(base + 16) = -pc.ct_B[set] * v1; PGS(base + 17) = -pc.ct_B[set] * v2;
    PGS(base + 18) = 1 / Rn;
    PGS(base + 19) = pc.ct_fric[set] * sqrt(1 / fmax(1e-15, pc.impratio));
  }
  // joint-limit rows of the arm (mj_instantiateLimit): at most one side per joint can be inside its margin
  LimitRow lim[NDOF];
  for (int k = 0; k < NDOF; k++) {
    double dlo = st.q[k] - c.jnt_range[k][0], dhi = c.jnt_range[k][1] - st.q[k];
    double sign = 0, dist = 0;
    if (dlo < c.lim_margin[k]) { sign = 1; dist = dlo; }
    else if (dhi < c.lim_margin[k]) { sign = -1; dist = dhi; }
    lim[k].sign = sign; lim[k].D = 0; lim[k].aref = 0;
    if (sign != 0) {
      double imp = impedance(c.lim_solimp[k], dist - c.lim_margin[k]);
      lim[k].D = 1 / fmax(1e-15, (1 - imp) / imp * c.dof_invweight0[k]);
      lim[k].aref = -c.lim_B[k] * (sign * st.v[k]) - c.lim_K[k] * imp * (dist - c.lim_margin[k]);
    }
  }
  bool any_lim = false;
  for (int k = 0; k < NDOF; k++) any_lim = any_lim || lim[k].sign != 0;
  // ---- Newton
  const bool env_bb = wave_any(has_bb), env_r1 = wave_any(has_rod[0]), env_r2 = wave_any(has_rod[1]);
  const double impr = pc.impratio;
  double x[PUSH_NV];
  if (ncon == 0 && !any_lim) {
    for (int k = 0; k < PUSH_NV; k++) x[k] = PGS(PG_A0 + k);
  } else {
    if (st.flags & PF_WARM_VALID) for (int k = 0; k < PUSH_NV; k++) x[k] = ps.warm[k];
    else for (int k = 0; k < PUSH_NV; k++) x[k] = PGS(PG_A0 + k);
    bool converged = false;
    for (int it = 0; it < PUSH_MAXIT && !converged; it++) {
      // gradient and Hessian at x
      double grad[PUSH_NV];
      {
        double dx[PUSH_NV];
        for (int k = 0; k < PUSH_NV; k++) dx[k] = x[k] - PGS(PG_A0 + k);
        for (int b = 0; b < PUSH_NB; b++) for (int k = 0; k < 6; k++) grad[6 * b + k] = (k < 3 ? pc.box_mass : pc.box_inertia) * dx[6 * b + k];
        for (int i = 0; i < NDOF; i++) {
          double s = 0;
          for (int k = 0; k < NDOF; k++) s += PGS(PG_M + (i >= k ? tri(i, k) : tri(k, i))) * dx[PUSH_ARM0 + k];
          grad[PUSH_ARM0 + i] = s;
        }
      }
      for (int i = 0; i < PUSH_NH; i++) PHS(i) = 0;
      for (int b = 0; b < PUSH_NB; b++) for (int k = 0; k < 6; k++) PHS(tri(6 * b + k, 6 * b + k)) = k < 3 ? pc.box_mass : pc.box_inertia;
      for (int i = 0; i < NDOF; i++) for (int k = 0; k <= i; k++) PHS(tri(PUSH_ARM0 + i, PUSH_ARM0 + k)) = PGS(PG_M + tri(i, k));
      for (int k = 0; k < NDOF; k++) if (lim[k].sign != 0) {
        double jar = lim[k].sign * x[PUSH_ARM0 + k] - lim[k].aref;
        if (jar < 0) { grad[PUSH_ARM0 + k] += lim[k].sign * lim[k].D * jar; PHS(tri(PUSH_ARM0 + k, PUSH_ARM0 + k)) += lim[k].D; }
      }
      for (int ci = 0; ci < ncon; ci++) {
        int base = PG_CON + ci * PREC;
        SRow rows[3];
        contact_rows(sc, ci, Rb, ps.box, rows);
        double jar[3], force[3], Hc[9];
        for (int r = 0; r < 3; r++) { jar[r] = srow_dot(rows[r], x) - PGS(base + 15 + r); PGS(base + 20 + r) = jar[r]; }
        double Dn = PGS(base + 18), mu = PGS(base + 19), fric = pc.ct_fric[(int)PGS(base + 13) == CK_SLAB ? 0 : 1];
        cone_eval(jar, Dn, Dn * impr, mu, fric, force, Hc);
        for (int r = 0; r < 3; r++) {
          if (force[r] == 0) continue;
          for (int k = 0; k < rows[r].n1; k++) grad[rows[r].o1 + k] -= rows[r].v1[k] * force[r];
          for (int k = 0; k < rows[r].n2; k++) grad[rows[r].o2 + k] -= rows[r].v2[k] * force[r];
        }
        if (Hc[0] == 0 && Hc[4] == 0) continue;
        // H += J' Hc J over the (up to two) dof blocks of this contact
        const int o1 = rows[0].o1, n1 = rows[0].n1, o2 = rows[0].o2, n2 = rows[0].n2;
        for (int a = 0; a < n1 + n2; a++) {
          int ia = a < n1 ? o1 + a : o2 + a - n1;
          double ja[3], ta[3];
          for (int r = 0; r < 3; r++) ja[r] = a < n1 ? rows[r].v1[a] : rows[r].v2[a - n1];
          for (int r = 0; r < 3; r++) ta[r] = Hc[3 * r] * ja[0] + Hc[3 * r + 1] * ja[1] + Hc[3 * r + 2] * ja[2];
          for (int bq = 0; bq <= a; bq++) {
            int ib = bq < n1 ? o1 + bq : o2 + bq - n1;
            double s = 0;
            for (int r = 0; r < 3; r++) s += ta[r] * (bq < n1 ? rows[r].v1[bq] : rows[r].v2[bq - n1]);
            PHS(ia >= ib ? tri(ia, ib) : tri(ib, ia)) += s;
          }
        }
      }
      // direction p = -H^-1 grad (kept in grad[])
      double gmax = 0;
      for (int k = 0; k < PUSH_NV; k++) gmax = fmax(gmax, fabs(grad[k]));
      if (!sky_chol(sc, env_bb, env_r1, env_r2)) { st.flags |= F_SOLVER_FAIL; break; }
      double p[PUSH_NV];
      for (int k = 0; k < PUSH_NV; k++) p[k] = -grad[k];
      sky_solve(sc, env_bb, env_r1, env_r2, p);
      // quadratic part along p
      double pMp = 0, pMa = 0;
      for (int b = 0; b < PUSH_NB; b++) for (int k = 0; k < 6; k++) {
        double mm = k < 3 ? pc.box_mass : pc.box_inertia, pk = p[6 * b + k];
        pMp += mm * pk * pk; pMa += mm * pk * (x[6 * b + k] - PGS(PG_A0 + 6 * b + k));
      }
      for (int i = 0; i < NDOF; i++) {
        double s = 0, sa = 0;
        for (int k = 0; k < NDOF; k++) { double m_ik = PGS(PG_M + (i >= k ? tri(i, k) : tri(k, i))); s += m_ik * p[PUSH_ARM0 + k]; sa += m_ik * (x[PUSH_ARM0 + k] - PGS(PG_A0 + PUSH_ARM0 + k)); }
        pMp += p[PUSH_ARM0 + i] * s; pMa += p[PUSH_ARM0 + i] * sa;
      }
      for (int ci = 0; ci < ncon; ci++) {
        int base = PG_CON + ci * PREC;
        SRow rows[3];
        contact_rows(sc, ci, Rb, ps.box, rows);
        for (int r = 0; r < 3; r++) PGS(base + 23 + r) = srow_dot(rows[r], p);
      }
      // exact line search: root of phi'(alpha) by safeguarded Newton (phi is convex, C1)
      double alpha = 1, lo = 0, hi = -1, best = 1;
      for (int ls = 0; ls < 40; ls++) {
        double d1 = pMa + alpha * pMp, d2 = pMp;
        for (int k = 0; k < NDOF; k++) if (lim[k].sign != 0) {
          double jp = lim[k].sign * p[PUSH_ARM0 + k], jar = lim[k].sign * x[PUSH_ARM0 + k] - lim[k].aref + alpha * jp;
          if (jar < 0) { d1 += lim[k].D * jar * jp; d2 += lim[k].D * jp * jp; }
        }
        for (int ci = 0; ci < ncon; ci++) {
          int base = PG_CON + ci * PREC;
          double jp[3] = {PGS(base + 23), PGS(base + 24), PGS(base + 25)};
          double jt[3] = {PGS(base + 20) + alpha * jp[0], PGS(base + 21) + alpha * jp[1], PGS(base + 22) + alpha * jp[2]}, ft[3], Hc[9];
          double Dn = PGS(base + 18), mu = PGS(base + 19), fric = pc.ct_fric[(int)PGS(base + 13) == CK_SLAB ? 0 : 1];
          cone_eval(jt, Dn, Dn * impr, mu, fric, ft, Hc);
          for (int r = 0; r < 3; r++) { d1 -= ft[r] * jp[r]; for (int q = 0; q < 3; q++) d2 += jp[r] * Hc[3 * r + q] * jp[q]; }
        }
        best = alpha;
        if (fabs(d1) < 1e-14 * fmax(1.0, fabs(pMa))) break;
        if (d1 < 0) lo = alpha; else hi = alpha;
        double na = alpha - d1 / d2;
        if (hi >= 0 && !(na > lo && na < hi)) na = 0.5 * (lo + hi);
        if (hi < 0 && na <= lo) na = 2 * lo + 1;
        if (na == alpha) break;
        alpha = na;
      }
      double smax = 0, xmax = 0;
      for (int k = 0; k < PUSH_NV; k++) { double dxk = best * p[k]; x[k] += dxk; smax = fmax(smax, fabs(dxk)); xmax = fmax(xmax, fabs(x[k])); }
      if (smax <= 1e-12 * (1 + xmax)) converged = true;
      (void)gmax;
    }
    if (!converged) st.flags |= F_SOLVER_FAIL;
    for (int k = 0; k < PUSH_NV; k++) ps.warm[k] = x[k];
    st.flags |= PF_WARM_VALID;
  }
  // ---- semi-implicit Euler.  Arm: (M + h B) qacc = M x (= qfrc_smooth + qfrc_constraint at the optimum), B on the fingers
  {
    double Md[45], rhs[NDOF];
    for (int i = 0; i < 45; i++) Md[i] = PGS(PG_M + i);
    for (int i = 0; i < NDOF; i++) {
      double s = 0;
      for (int k = 0; k < NDOF; k++) s += Md[i >= k ? tri(i, k) : tri(k, i)] * x[PUSH_ARM0 + k];
      rhs[i] = s;
    }
    Md[tri(7, 7)] += h * c.f_damping[0]; Md[tri(8, 8)] += h * c.f_damping[1];
    double L[45], d[NDOF], id[NDOF];
    if (!ldl9(Md, L, d, id)) st.flags |= F_SOLVER_FAIL;
    ldl9_solve(L, id, rhs);
    for (int k = 0; k < NDOF; k++) { st.v[k] += h * rhs[k]; st.q[k] += h * st.v[k]; }
  }
  for (int b = 0; b < PUSH_NB; b++) {
    BoxState& bx = ps.box[b];
    for (int k = 0; k < 6; k++) bx.vel[k] += h * x[6 * b + k];
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
      for (int k = 0; k < 4; k++) bx.quat[k] = r[k] / nn;
    }
  }
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
D3IL_HD void push_dists(const PushConsts& pc, const PushState& ps, double* d) {   // rr rg gr gg
  for (int b = 0; b < 2; b++) for (int t = 0; t < 2; t++) {
    double dx = ps.box[b].pos[0] - pc.target[t][0], dy = ps.box[b].pos[1] - pc.target[t][1], dz = ps.box[b].pos[2] - pc.target[t][2];
    d[2 * b + t] = sqrt(dx * dx + dy * dy + dz * dz);
  }
}
D3IL_HD bool push_success(const PushConsts& pc, const PushState& ps) {   // pushing.py:440-459
  double d[4]; push_dists(pc, ps, d);
  return (d[0] <= pc.min_dist && d[3] <= pc.min_dist) || (d[1] <= pc.min_dist && d[2] <= pc.min_dist);
}
D3IL_HD void push_obs(const PushState& ps, float* obs) {   // pushing.py:255-280
  obs[0] = (float)ps.arm.tcp[0]; obs[1] = (float)ps.arm.tcp[1];
  for (int b = 0; b < 2; b++) { obs[2 + 3 * b] = (float)ps.box[b].pos[0]; obs[3 + 3 * b] = (float)ps.box[b].pos[1]; obs[4 + 3 * b] = (float)push_tan_yaw(ps.box[b].quat); }
}
// before the physics of a step: obs, reward, done (gym_env_wrapper.py:88-90,124-137)
D3IL_HD void push_step_begin(const PushConsts& pc, PushState& ps, float* obs, double* reward, unsigned char* done, int max_steps) {
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
D3IL_HD void push_step_end(const PushConsts& pc, PushState& ps, double* mean_distance) {
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
// joint PD on the set-point + finger PD + one physics sub-step (Scene.next_step after the IK update)
template <class C>
D3IL_HD void push_control_and_physics(const C& c, const PushConsts& pc, PushState& ps, const PushScratch& sc, const double* q_des, const double* qd_des,
                                      double set_width, bool grasp) {
  EnvState& st = ps.arm;
  double tau[NARM], ff[NFING];
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
  push_physics_substep(c, pc, ps, sc, tau, ff);
}

// Block_Push_Env.reset(random=False, context) (pushing.py:461-483): scene.reset, beam to init_qpos, context written into
// the cubes' qpos (z = 0, pushing.py:99-113), one PD-hold sub-step.  ctx = 2 x (pos3, quat4).
template <class C>
D3IL_HD void push_env_reset(const C& c, const PushConsts& pc, PushState& ps, const PushScratch& sc, const double* init_qpos, const double* ctx, float* obs) {
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
  for (int k = 0; k < PUSH_NV; k++) ps.warm[k] = 0;
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
D3IL_HD void push_env_step(const C& c, const PushConsts& pc, PushState& ps, const PushScratch& sc, const double* action, float* obs, double* reward,
                           unsigned char* done, double* mean_distance, int n_substeps, int max_steps) {
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
