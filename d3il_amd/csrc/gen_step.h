// gen_step.h - generic (memory-resident) sub-step for "Panda arm + rod + NB free cubes + NS static boxes": the engine the
// Sorting task (4 cubes, 2 table slabs, 8 bin walls, platform) runs on until it gets specialised fast paths like Pushing's.
//
// Everything that varies in size lives in the per-lane scratch areas (contact records, vectors in HBM; the dense
// (6 NB + 9)^2 Hessian in the LDS area), so the code is loop based and register-light:
//   collision   every cube against every static box (sphere pre-test, then box_box in the oracle's geom order), cube
//               pairs, rod against every cube;
//   constraints 9 joint-limit rows + condim-3 elliptic cones, contact parameters mixed per pair (priority: the platform's
//               own solref / solimp / friction, sorting platform.xml);
//   solve       primal Newton with safeguarded exact line search on the full system, dense LDL^T in the h area;
//   integrate   semi-implicit Euler (arm with implicit finger damping, cubes with quaternion integration).
// Same numerics as the oracle's generic engine up to solver tolerance; compared with it in tests/test_sorting_host.py.
#pragma once
#include "push_step.h"

namespace d3il {

constexpr int GEN_MAXNB = 4, GEN_MAXNS = 12, GEN_MAXCON = 72, GEN_MAXSET = GEN_MAXNS + 2;
constexpr int GEN_MAXNV = 6 * GEN_MAXNB + NDOF;     // 33
constexpr int GEN_NH = GEN_MAXNV * (GEN_MAXNV + 1) / 2;   // 561
constexpr int GEN_LANES = 32;           // environments per workgroup: 32 x 561 doubles of LDS for the Hessians

struct GenConsts {
  int nb, ns, set_bb, set_rod;
  double box_half[3], box_mass, box_inertia, box_invw_t;
  double st_c[GEN_MAXNS][3], st_h[GEN_MAXNS][3], st_R[GEN_MAXNS][9];
  int st_first[GEN_MAXNS];        // 1: the static geom precedes the cube geoms in the model (it is geom 1 of the pair)
  double ct_K[GEN_MAXSET], ct_B[GEN_MAXSET], ct_solimp[GEN_MAXSET][5], ct_fric[GEN_MAXSET];   // set s < ns: static s <-> cube
  double impratio, rod_invw;
  double ws_lo[2], ws_hi[2];      // modelled workspace of the cube centres (x, y)
  double absent[7];               // pose reported for boxes the model does not have (body id -1)
};

// The w area of the scratch views is the environment's object block of the state buffer (rows 42 ..): the cubes' pos[3] quat[4]
// vel[6], then the solver's warm start [nv], then the two task words (stored as doubles).  The cubes stay there for the
// whole step - only the arm lives in registers.
#define GBX(b, k) PWS(13 * (b) + (k))
#define GWARM(k) PWS(13 * gc.nb + (k))
#define GTASK(k) PWS(13 * gc.nb + 6 * gc.nb + NDOF + (k))
D3IL_HD constexpr int gen_state_rows(int nb) { return 42 + 13 * nb + 6 * nb + NDOF + 2; }

// g-area layout (doubles per lane)
constexpr int GG_M = 0, GG_A0 = 45, GG_X = GG_A0 + GEN_MAXNV, GG_P = GG_X + GEN_MAXNV, GG_G = GG_P + GEN_MAXNV, GG_VEL = GG_G + GEN_MAXNV;
constexpr int GG_R = GG_VEL + GEN_MAXNV, GG_POS = GG_R + 9 * GEN_MAXNB, GG_LIM = GG_POS + 3 * GEN_MAXNB, GG_JA = GG_LIM + 27;
constexpr int GG_CON = GG_JA + 21 * GEN_MAXNB;
constexpr int GREC = 28;   // pos[3] frame[9] dist kind a b | aref[3] Dn mu fric | jar[3] jp[3]
constexpr int GG_SIZE = GG_CON + GEN_MAXCON * GREC;
enum { GK_STATIC = 0, GK_BOXBOX = 1, GK_ROD = 2 };

struct GRow { int o1, o2, n2; double v1[6], v2[7]; };

// rows of contact ci over the solver dofs [cube0 .. cube(nb-1) | arm]; the contact normal points from geom 1 to geom 2 and
// the row is J(body 2) - J(body 1)
D3IL_HD void gen_rows(const GenConsts& gc, const PushScratch& sc, int ci, GRow* rows) {
  const int base = GG_CON + ci * GREC, arm0 = 6 * gc.nb;
  double p[3] = {PGS(base), PGS(base + 1), PGS(base + 2)};
  const int kind = (int)PGS(base + 13), a = (int)PGS(base + 14), b = (int)PGS(base + 15);
  const int c1 = kind == GK_STATIC ? b : a;        // first cube of the row (static: b = cube, a = static index)
  double R[9], r[3];
  for (int k = 0; k < 9; k++) R[k] = PGS(GG_R + 9 * c1 + k);
  for (int k = 0; k < 3; k++) r[k] = p[k] - PGS(GG_POS + 3 * c1 + k);
  for (int rr = 0; rr < 3; rr++) {
    double f[3] = {PGS(base + 3 + 3 * rr), PGS(base + 4 + 3 * rr), PGS(base + 5 + 3 * rr)};
    GRow& s = rows[rr];
    box_row_r(R, r, f, s.v1);
    for (int k = 0; k < 7; k++) s.v2[k] = 0;
    s.o1 = 6 * c1; s.o2 = 0; s.n2 = 0;
    if (kind == GK_STATIC) {
      if (!gc.st_first[a]) for (int k = 0; k < 6; k++) s.v1[k] = -s.v1[k];       // cube is geom 1
    } else if (kind == GK_BOXBOX) {
      for (int k = 0; k < 6; k++) s.v1[k] = -s.v1[k];
      double R2[9], r2[3], t[6];
      for (int k = 0; k < 9; k++) R2[k] = PGS(GG_R + 9 * b + k);
      for (int k = 0; k < 3; k++) r2[k] = p[k] - PGS(GG_POS + 3 * b + k);
      box_row_r(R2, r2, f, t);
      s.o2 = 6 * b; s.n2 = 6;
      for (int k = 0; k < 6; k++) s.v2[k] = t[k];
    } else {
      for (int k = 0; k < 6; k++) s.v1[k] = -s.v1[k];
      s.o2 = arm0; s.n2 = 7;
      for (int k = 0; k < 7; k++) s.v2[k] = PGS(GG_JA + 21 * a + 7 * rr + k);
    }
  }
}
D3IL_HD double grow_dot(const PushScratch& sc, const GRow& s, int vec) {
  double acc = 0;
  for (int k = 0; k < 6; k++) acc += s.v1[k] * PGS(vec + s.o1 + k);
  for (int k = 0; k < s.n2; k++) acc += s.v2[k] * PGS(vec + s.o2 + k);
  return acc;
}
D3IL_HD double gen_M(const GenConsts& gc, const PushScratch& sc, int i, int k) {
  const int arm0 = 6 * gc.nb;
  if (i < arm0 || k < arm0) return i == k ? ((i % 6) < 3 ? gc.box_mass : gc.box_inertia) : 0.0;
  int a = i - arm0, b = k - arm0;
  return PGS(GG_M + (a >= b ? tri(a, b) : tri(b, a)));
}
// H (h area, packed lower, order nv) -> L L^T in place; solve in place on the g vector at `vec`
#if defined(__HIP_DEVICE_COMPILE__)
#define GEN_HS GEN_LANES
#else
#define GEN_HS 1
#endif
#define GHS(i) sc.h[(i) * GEN_HS]
D3IL_HD bool gen_chol(const PushScratch& sc, int nv) {
  bool ok = true;
  for (int i = 0; i < nv; i++)
    for (int j = 0; j <= i; j++) {
      double s = GHS(tri(i, j));
      for (int k = 0; k < j; k++) s -= GHS(tri(i, k)) * GHS(tri(j, k));
      if (i == j) { if (!(s > 0)) { ok = false; s = 1; } GHS(tri(i, i)) = sqrt(s); }
      else GHS(tri(i, j)) = s / GHS(tri(j, j));
    }
  return ok;
}
D3IL_HD void gen_chol_solve(const PushScratch& sc, int nv, int vec) {
  for (int i = 0; i < nv; i++) {
    double s = PGS(vec + i);
    for (int k = 0; k < i; k++) s -= GHS(tri(i, k)) * PGS(vec + k);
    PGS(vec + i) = s / GHS(tri(i, i));
  }
  for (int i = nv - 1; i >= 0; i--) {
    double xi = PGS(vec + i) / GHS(tri(i, i));
    PGS(vec + i) = xi;
    for (int k = 0; k < i; k++) PGS(vec + k) -= GHS(tri(i, k)) * xi;
  }
}

// contact collection; returns the count (contacts beyond GEN_MAXCON are dropped and flagged)
D3IL_NOINLINE inline int gen_collect(const GenConsts& gc, const PushScratch& sc, const double* rodc, const double* rodu, double rod_r, double rod_h, unsigned* flags) {
  int ncon = 0;
  const double rcirc = sqrt(gc.box_half[0] * gc.box_half[0] + gc.box_half[1] * gc.box_half[1] + gc.box_half[2] * gc.box_half[2]);
  auto put = [&](const double* rec, double nsign, int kind, int a, int b, int set) {
    if (ncon >= GEN_MAXCON) { *flags |= PF_CON_OVERFLOW; return; }
    int base = GG_CON + ncon * GREC;
    double n[3] = {nsign * rec[4], nsign * rec[5], nsign * rec[6]}, t1[3], t2[3];
    make_frame(n, t1, t2);
    for (int k = 0; k < 3; k++) { PGS(base + k) = rec[1 + k]; PGS(base + 3 + k) = n[k]; PGS(base + 6 + k) = t1[k]; PGS(base + 9 + k) = t2[k]; }
    PGS(base + 12) = rec[0]; PGS(base + 13) = kind; PGS(base + 14) = a; PGS(base + 15) = b; PGS(base + 21) = set;
    ncon++;
  };
  double rec[8][7];
  for (int c = 0; c < gc.nb; c++) {
    double pc[3], Rc[9];
    for (int k = 0; k < 3; k++) pc[k] = PGS(GG_POS + 3 * c + k);
    for (int k = 0; k < 9; k++) Rc[k] = PGS(GG_R + 9 * c + k);
    for (int s = 0; s < gc.ns; s++) {
      // sphere against the static box (in its frame): cheap exact rejection
      double d2 = 0;
      for (int i = 0; i < 3; i++) {
        double x = (pc[0] - gc.st_c[s][0]) * gc.st_R[s][i] + (pc[1] - gc.st_c[s][1]) * gc.st_R[s][3 + i] + (pc[2] - gc.st_c[s][2]) * gc.st_R[s][6 + i];
        double e = fabs(x) - gc.st_h[s][i];
        if (e > 0) d2 += e * e;
      }
      if (d2 > rcirc * rcirc) continue;
      int n = gc.st_first[s] ? box_box(gc.st_c[s], gc.st_R[s], gc.st_h[s], pc, Rc, gc.box_half, 0.0, rec, 8)
                             : box_box(pc, Rc, gc.box_half, gc.st_c[s], gc.st_R[s], gc.st_h[s], 0.0, rec, 8);
      for (int i = 0; i < n; i++) put(rec[i], 1.0, GK_STATIC, s, c, s);
    }
  }
  for (int c = 0; c < gc.nb; c++) for (int d = c + 1; d < gc.nb; d++) {
    double pc[3], pd[3], Rc[9], Rd[9], dd = 0;
    for (int k = 0; k < 3; k++) { pc[k] = PGS(GG_POS + 3 * c + k); pd[k] = PGS(GG_POS + 3 * d + k); dd += (pc[k] - pd[k]) * (pc[k] - pd[k]); }
    if (dd > 4 * rcirc * rcirc) continue;
    for (int k = 0; k < 9; k++) { Rc[k] = PGS(GG_R + 9 * c + k); Rd[k] = PGS(GG_R + 9 * d + k); }
    int n = box_box(pc, Rc, gc.box_half, pd, Rd, gc.box_half, 0.0, rec, 8);
    for (int i = 0; i < n; i++) put(rec[i], 1.0, GK_BOXBOX, c, d, gc.set_bb);
  }
  for (int c = 0; c < gc.nb; c++) {
    double pc[3], Rc[9], r1[7];
    for (int k = 0; k < 3; k++) pc[k] = PGS(GG_POS + 3 * c + k);
    for (int k = 0; k < 9; k++) Rc[k] = PGS(GG_R + 9 * c + k);
    double w[3] = {pc[0] - rodc[0], pc[1] - rodc[1], pc[2] - rodc[2]};
    double t = clampd(dot3(w, rodu), -rod_h, rod_h);
    double e[3] = {w[0] - t * rodu[0], w[1] - t * rodu[1], w[2] - t * rodu[2]};
    if (dot3(e, e) >= (rcirc + rod_r) * (rcirc + rod_r)) continue;
    if (cyl_box(rodc, rodu, rod_r, rod_h, pc, Rc, gc.box_half, 0.0, r1)) put(r1, 1.0, GK_ROD, c, 0, gc.set_rod);   // cube is geom 1: normal cube -> rod
  }
  return ncon;
}

// Newton solve on the collected system; x in / out at GG_X.  Returns false when it did not converge.
D3IL_NOINLINE inline bool gen_solve(const GenConsts& gc, const PushScratch& sc, int ncon) {
  const int arm0 = 6 * gc.nb, nv = arm0 + NDOF;
  const double impr = gc.impratio;
  for (int ci = 0; ci < ncon; ci++) {   // reference acceleration and regularisation
    int base = GG_CON + ci * GREC, kind = (int)PGS(base + 13), set = (int)PGS(base + 21);
    double dist = PGS(base + 12);
    double imp = impedance(gc.ct_solimp[set], dist);
    double invw = kind == GK_STATIC ? gc.box_invw_t : (kind == GK_BOXBOX ? 2 * gc.box_invw_t : gc.box_invw_t + gc.rod_invw);
    GRow rows[3];
    gen_rows(gc, sc, ci, rows);
    double v0 = grow_dot(sc, rows[0], GG_VEL), v1 = grow_dot(sc, rows[1], GG_VEL), v2 = grow_dot(sc, rows[2], GG_VEL);
    PGS(base + 16) = -gc.ct_B[set] * v0 - gc.ct_K[set] * imp * dist;
    PGS(base + 17) = -gc.ct_B[set] * v1; PGS(base + 18) = -gc.ct_B[set] * v2;
    PGS(base + 19) = 1 / fmax(1e-15, (1 - imp) / imp * invw);
    PGS(base + 20) = gc.ct_fric[set] * sqrt(1 / fmax(1e-15, impr));
  }
  bool converged = false;
  for (int it = 0; it < 60 && !converged; it++) {
    for (int i = 0; i < nv; i++) {
      double s = 0;
      if (i < arm0) s = gen_M(gc, sc, i, i) * (PGS(GG_X + i) - PGS(GG_A0 + i));
      else for (int k = arm0; k < nv; k++) s += gen_M(gc, sc, i, k) * (PGS(GG_X + k) - PGS(GG_A0 + k));
      PGS(GG_G + i) = s;
    }
    for (int i = 0; i < nv * (nv + 1) / 2; i++) GHS(i) = 0;
    for (int i = 0; i < arm0; i++) GHS(tri(i, i)) = gen_M(gc, sc, i, i);
    for (int i = 0; i < NDOF; i++) for (int k = 0; k <= i; k++) GHS(tri(arm0 + i, arm0 + k)) = PGS(GG_M + tri(i, k));
    for (int k = 0; k < NDOF; k++) {
      double sign = PGS(GG_LIM + 3 * k), D = PGS(GG_LIM + 3 * k + 1), aref = PGS(GG_LIM + 3 * k + 2);
      if (sign != 0) {
        double jar = sign * PGS(GG_X + arm0 + k) - aref;
        if (jar < 0) { PGS(GG_G + arm0 + k) += sign * D * jar; GHS(tri(arm0 + k, arm0 + k)) += D; }
      }
    }
    for (int ci = 0; ci < ncon; ci++) {
      int base = GG_CON + ci * GREC;
      GRow rows[3];
      gen_rows(gc, sc, ci, rows);
      double jar[3], force[3], Hc[9];
      for (int r = 0; r < 3; r++) { jar[r] = grow_dot(sc, rows[r], GG_X) - PGS(base + 16 + r); PGS(base + 22 + r) = jar[r]; }
      double Dn = PGS(base + 19), mu = PGS(base + 20), fric = gc.ct_fric[(int)PGS(base + 21)];
      cone_eval(jar, Dn, Dn * impr, mu, fric, force, Hc);
      if (force[0] == 0 && force[1] == 0 && force[2] == 0) continue;
      const int o1 = rows[0].o1, o2 = rows[0].o2, n2 = rows[0].n2;
      for (int k = 0; k < 6; k++) PGS(GG_G + o1 + k) -= rows[0].v1[k] * force[0] + rows[1].v1[k] * force[1] + rows[2].v1[k] * force[2];
      for (int k = 0; k < n2; k++) PGS(GG_G + o2 + k) -= rows[0].v2[k] * force[0] + rows[1].v2[k] * force[1] + rows[2].v2[k] * force[2];
      for (int a = 0; a < 6; a++) {
        double ta[3];
        for (int r = 0; r < 3; r++) ta[r] = Hc[3 * r] * rows[0].v1[a] + Hc[3 * r + 1] * rows[1].v1[a] + Hc[3 * r + 2] * rows[2].v1[a];
        for (int b = 0; b <= a; b++) GHS(tri(o1 + a, o1 + b)) += ta[0] * rows[0].v1[b] + ta[1] * rows[1].v1[b] + ta[2] * rows[2].v1[b];
      }
      for (int a = 0; a < n2; a++) {
        double ta[3];
        for (int r = 0; r < 3; r++) ta[r] = Hc[3 * r] * rows[0].v2[a] + Hc[3 * r + 1] * rows[1].v2[a] + Hc[3 * r + 2] * rows[2].v2[a];
        // o2 > o1 for every contact kind (second cube index > first, arm after the cubes)
        for (int b = 0; b < 6; b++) GHS(tri(o2 + a, o1 + b)) += ta[0] * rows[0].v1[b] + ta[1] * rows[1].v1[b] + ta[2] * rows[2].v1[b];
        for (int b = 0; b <= a; b++) GHS(tri(o2 + a, o2 + b)) += ta[0] * rows[0].v2[b] + ta[1] * rows[1].v2[b] + ta[2] * rows[2].v2[b];
      }
    }
    {
      double gm = 0;
      for (int k = 0; k < nv; k++) gm = fmax(gm, fabs(PGS(GG_G + k)));
      if (gm <= PUSH_GRAD_TOL) { converged = true; break; }
    }
    if (!gen_chol(sc, nv)) return false;
    for (int k = 0; k < nv; k++) PGS(GG_P + k) = -PGS(GG_G + k);
    gen_chol_solve(sc, nv, GG_P);
    double pMp = 0, pMa = 0, gTp = 0;
    for (int i = 0; i < nv; i++) {
      double s = 0, sa = 0;
      if (i < arm0) { double mm = gen_M(gc, sc, i, i); s = mm * PGS(GG_P + i); sa = mm * (PGS(GG_X + i) - PGS(GG_A0 + i)); }
      else for (int k = arm0; k < nv; k++) { double mm = gen_M(gc, sc, i, k); s += mm * PGS(GG_P + k); sa += mm * (PGS(GG_X + k) - PGS(GG_A0 + k)); }
      pMp += PGS(GG_P + i) * s; pMa += PGS(GG_P + i) * sa; gTp += PGS(GG_G + i) * PGS(GG_P + i);
    }
    for (int ci = 0; ci < ncon; ci++) {
      int base = GG_CON + ci * GREC;
      GRow rows[3];
      gen_rows(gc, sc, ci, rows);
      for (int r = 0; r < 3; r++) PGS(base + 25 + r) = grow_dot(sc, rows[r], GG_P);
    }
    double alpha = 1, lo = 0, hi = -1, best = 1, wprev = 1e300;
    for (int ls = 0; ls < 50; ls++) {
      double d1 = pMa + alpha * pMp, d2 = pMp;
      for (int k = 0; k < NDOF; k++) {
        double sign = PGS(GG_LIM + 3 * k), D = PGS(GG_LIM + 3 * k + 1), aref = PGS(GG_LIM + 3 * k + 2);
        if (sign != 0) {
          double jp = sign * PGS(GG_P + arm0 + k), jar = sign * PGS(GG_X + arm0 + k) - aref + alpha * jp;
          if (jar < 0) { d1 += D * jar * jp; d2 += D * jp * jp; }
        }
      }
      for (int ci = 0; ci < ncon; ci++) {
        int base = GG_CON + ci * GREC;
        double jp[3] = {PGS(base + 25), PGS(base + 26), PGS(base + 27)};
        double jt[3] = {PGS(base + 22) + alpha * jp[0], PGS(base + 23) + alpha * jp[1], PGS(base + 24) + alpha * jp[2]}, ft[3], Hc[9];
        double Dn = PGS(base + 19), mu = PGS(base + 20), fric = gc.ct_fric[(int)PGS(base + 21)];
        cone_eval(jt, Dn, Dn * impr, mu, fric, ft, Hc);
        for (int r = 0; r < 3; r++) { d1 -= ft[r] * jp[r]; for (int q = 0; q < 3; q++) d2 += jp[r] * Hc[3 * r + q] * jp[q]; }
      }
      best = alpha;
      if (ls == 0 && d1 <= 0.1 * fabs(gTp)) break;
      if (fabs(d1) <= 1e-3 * d2 * alpha || fabs(d1) < 1e-14 * fmax(1.0, fabs(pMa))) break;
      if (d1 < 0) lo = alpha; else hi = alpha;
      double na = alpha - d1 * rcpd(d2);
      if (hi >= 0) {
        double wbr = hi - lo;
        bool slow = wbr > 0.5 * wprev;
        wprev = wbr;
        if (slow || !(na > lo && na < hi)) na = 0.5 * (lo + hi);
      } else if (na <= lo) na = 2 * lo + 1;
      if (na == alpha) break;
      alpha = na;
    }
    double smax = 0, xmax = 0;
    for (int k = 0; k < nv; k++) {
      double dxk = best * PGS(GG_P + k), xn = PGS(GG_X + k) + dxk;
      PGS(GG_X + k) = xn; smax = fmax(smax, fabs(dxk)); xmax = fmax(xmax, fabs(xn));
    }
    if (smax <= 1e-12 * (1 + xmax) || (best == 1.0 && smax <= 1e-6 * (1 + xmax))) converged = true;
  }
  return converged;
}

// one physics sub-step (mj_step) of arm + cubes; warm start / result of the solver at sc.w[0 .. nv)
template <class C>
D3IL_HD void gen_physics_substep(const C& c0, const GenConsts& gc, EnvState& st, const PushScratch& sc, const double* tau, const double* ffing) {
  D3IL_REFRESH(c0, c);
  const double h = c.timestep;
  const int arm0 = 6 * gc.nb, nv = arm0 + NDOF;
  DynOut dyn;
  dynamics(c0, st.q, st.v, dyn);
  double fs[NDOF];
  for (int k = 0; k < NARM; k++) fs[k] = clampd(tau[k] + st.bias[k], c.force_lo[k], c.force_hi[k]) - dyn.bias[k];
  for (int k = 0; k < NFING; k++) fs[NARM + k] = clampd(ffing[k], c.force_lo[NARM + k], c.force_hi[NARM + k]) - dyn.bias[NARM + k] - c.f_damping[k] * st.v[NARM + k];
  for (int k = 0; k < NARM; k++) st.bias[k] = dyn.bias[k];
  double rodc[3], rodu[3];
  {
    double t[3]; mulE(dyn.R7, c.tcp7, t);
    st.tcp[0] = dyn.p7[0] + t[0]; st.tcp[1] = dyn.p7[1] + t[1]; st.tcp[2] = dyn.p7[2] + t[2];
    mulE(dyn.R7, c.rod_c7, rodc); rodc[0] += dyn.p7[0]; rodc[1] += dyn.p7[1]; rodc[2] += dyn.p7[2];
    mulE(dyn.R7, c.rod_u7, rodu);
  }
  double L[45], d[NDOF], id[NDOF], a0[NDOF];
  if (!ldl9(dyn.M, L, d, id)) st.flags |= F_SOLVER_FAIL;
  for (int k = 0; k < NDOF; k++) a0[k] = fs[k];
  ldl9_solve(L, id, a0);
  // publish the system to the scratch area
  for (int i = 0; i < 45; i++) PGS(GG_M + i) = dyn.M[i];
  for (int b = 0; b < gc.nb; b++) {
    double R[9], q[4] = {GBX(b, 3), GBX(b, 4), GBX(b, 5), GBX(b, 6)};
    quat2mat(q, R);
    for (int k = 0; k < 9; k++) PGS(GG_R + 9 * b + k) = R[k];
    for (int k = 0; k < 3; k++) PGS(GG_POS + 3 * b + k) = GBX(b, k);
    for (int k = 0; k < 6; k++) { PGS(GG_VEL + 6 * b + k) = GBX(b, 7 + k); PGS(GG_A0 + 6 * b + k) = k < 3 ? c.gravity[k] : 0.0; }
    if (GBX(b, 0) < gc.ws_lo[0] || GBX(b, 0) > gc.ws_hi[0] || GBX(b, 1) < gc.ws_lo[1] || GBX(b, 1) > gc.ws_hi[1]) st.flags |= PF_OFF_TABLE;
  }
  for (int k = 0; k < NDOF; k++) { PGS(GG_VEL + arm0 + k) = st.v[k]; PGS(GG_A0 + arm0 + k) = a0[k]; }
  bool any_lim = false;
  for (int k = 0; k < NDOF; k++) {
    double dlo = st.q[k] - c.jnt_range[k][0], dhi = c.jnt_range[k][1] - st.q[k];
    double sign = 0, dist = 0, D = 0, aref = 0;
    if (dlo < c.lim_margin[k]) { sign = 1; dist = dlo; }
    else if (dhi < c.lim_margin[k]) { sign = -1; dist = dhi; }
    if (sign != 0) {
      double imp = impedance(c.lim_solimp[k], dist - c.lim_margin[k]);
      D = 1 / fmax(1e-15, (1 - imp) / imp * c.dof_invweight0[k]);
      aref = -c.lim_B[k] * (sign * st.v[k]) - c.lim_K[k] * imp * (dist - c.lim_margin[k]);
      any_lim = true;
    }
    PGS(GG_LIM + 3 * k) = sign; PGS(GG_LIM + 3 * k + 1) = D; PGS(GG_LIM + 3 * k + 2) = aref;
  }
  unsigned cfl = 0;
  int ncon = gen_collect(gc, sc, rodc, rodu, c.rod_r, c.rod_h, &cfl);
  st.flags |= cfl;
  {   // arm Jacobian rows of the rod contacts
    bool any_rod = false;
    for (int ci = 0; ci < ncon; ci++) any_rod = any_rod || (int)PGS(GG_CON + ci * GREC + 13) == GK_ROD;
    if (any_rod) {
      double R7[9], p7[3], ax[NARM][3], og[NARM][3];
      world_chain(c0, dyn.sn, dyn.cs, R7, p7, ax, og);
      for (int ci = 0; ci < ncon; ci++) {
        int base = GG_CON + ci * GREC;
        if ((int)PGS(base + 13) != GK_ROD) continue;
        int b = (int)PGS(base + 14);
        double p[3] = {PGS(base), PGS(base + 1), PGS(base + 2)};
        for (int k = 0; k < NARM; k++) {
          double dd[3] = {p[0] - og[k][0], p[1] - og[k][1], p[2] - og[k][2]}, col[3];
          cross3(ax[k], dd, col);
          for (int r = 0; r < 3; r++) PGS(GG_JA + 21 * b + 7 * r + k) = col[0] * PGS(base + 3 + 3 * r) + col[1] * PGS(base + 4 + 3 * r) + col[2] * PGS(base + 5 + 3 * r);
        }
      }
    }
  }
  if (ncon == 0 && !any_lim) {
    for (int k = 0; k < nv; k++) PGS(GG_X + k) = PGS(GG_A0 + k);
  } else {
    if (st.flags & PF_WARM_VALID) for (int k = 0; k < nv; k++) PGS(GG_X + k) = GWARM(k);
    else for (int k = 0; k < nv; k++) PGS(GG_X + k) = PGS(GG_A0 + k);
    if (!gen_solve(gc, sc, ncon)) st.flags |= F_SOLVER_FAIL;
  }
  for (int k = 0; k < nv; k++) GWARM(k) = PGS(GG_X + k);
  st.flags |= PF_WARM_VALID;
  // arm: (M + h B) qacc = M x, B on the fingers
  {
    double xa[NDOF], rhs[NDOF];
    for (int k = 0; k < NDOF; k++) xa[k] = PGS(GG_X + arm0 + k);
    symv9(dyn.M, xa, rhs);
    double hb0 = h * c.f_damping[0], hb1 = h * c.f_damping[1];
    double l87 = L[tri(8, 7)];
    double S11 = d[8] + l87 * l87 * d[7];
    double d7n = d[7] + hb0, i7 = rcpd(d7n);
    double l87n = l87 * d[7] * i7;
    double d8n = S11 + hb1 - l87n * l87n * d7n;
    d[7] = d7n; d[8] = d8n; id[7] = i7; id[8] = rcpd(d8n); L[tri(8, 7)] = l87n;
    ldl9_solve(L, id, rhs);
    for (int k = 0; k < NDOF; k++) { st.v[k] += h * rhs[k]; st.q[k] += h * st.v[k]; }
  }
  for (int b = 0; b < gc.nb; b++) {
    double xb[6];
    BoxState bx;
    for (int k = 0; k < 6; k++) xb[k] = PGS(GG_X + 6 * b + k);
    for (int k = 0; k < 3; k++) bx.pos[k] = GBX(b, k);
    for (int k = 0; k < 4; k++) bx.quat[k] = GBX(b, 3 + k);
    for (int k = 0; k < 6; k++) bx.vel[k] = GBX(b, 7 + k);
    cube_integrate(bx, xb, h);
    for (int k = 0; k < 3; k++) GBX(b, k) = bx.pos[k];
    for (int k = 0; k < 4; k++) GBX(b, 3 + k) = bx.quat[k];
    for (int k = 0; k < 6; k++) GBX(b, 7 + k) = bx.vel[k];
  }
}

// ------------------------------------------------------------------------------------------------ Sorting task (sorting.py)
// Task state of Sorting_Env in two words: word 0 = mode[6] (2 bits each: value + 1) | mode_step << 12;  word 1 = min_inds[6]
// (3 bits each).  sorting.py:405-411 (reset), :460-507 (check_mode)
constexpr int GEN_SORT_OBS = 20;      // 2 + 3 * 6 at most (num_boxes = 6); Sorting-4 uses 14
D3IL_HD unsigned sort_word0_reset() { return 0; }
D3IL_HD void sort_collect(const GenConsts& gc, const PushScratch& sc, double (*box)[7]) {
  const int nr = gc.nb / 2;
  for (int i = 0; i < 6; i++) {
    const int cidx = i / 3, k = i % 3;
    if (k < nr) {
      for (int j = 0; j < 7; j++) box[i][j] = GBX(cidx * nr + k, j);
    } else for (int j = 0; j < 7; j++) box[i][j] = gc.absent[j];   // body id -1: the model's last body (MjScene.py:233-247)
  }
}
D3IL_HD bool sort_in_bin(const double* b, bool red) {
  return red ? (b[0] > 0.3 && b[0] < 0.5 && b[1] > 0.22 && b[1] < 0.41) : (b[0] > 0.525 && b[0] < 0.725 && b[1] > 0.22 && b[1] < 0.41);
}
// Sorting_Env.get_observation (sorting.py:308-390) and _check_early_termination (:513-543)
D3IL_HD bool sort_obs_success(const GenConsts& gc, const double (*box)[7], const double* tcp, float* obs) {
  const int nr = gc.nb / 2;
  int k = 0;
  obs[k++] = (float)tcp[0]; obs[k++] = (float)tcp[1];
  bool ok = true;
  for (int cidx = 0; cidx < 2; cidx++) for (int i = 0; i < nr; i++) {
    const double* b = box[3 * cidx + i];
    obs[k++] = (float)b[0]; obs[k++] = (float)b[1]; obs[k++] = (float)push_tan_yaw(b + 3);
    ok = ok && sort_in_bin(b, cidx == 0);
  }
  return ok;
}
// check_mode + decode_mode: int(np.packbits(mode[:num_boxes])[0]) - every non-zero entry (also -1) is a set bit, MSB first
D3IL_HD int sort_check_mode(const GenConsts& gc, unsigned* task, const double (*box)[7]) {
  int mode_step = (int)(task[0] >> 12) & 7;
  if (mode_step <= 5) {
    double dists[6];
    for (int i = 0; i < 6; i++) {
      double tx = i < 3 ? 0.4 : 0.625, ty = 0.32, dx = box[i][0] - tx, dy = box[i][1] - ty;
      dists[i] = sqrt(dx * dx + dy * dy);
    }
    for (int i = 0; i < mode_step; i++) dists[(task[1] >> (3 * i)) & 7] = 100000;
    int mi = 0;
    for (int i = 1; i < 6; i++) if (dists[i] < dists[mi]) mi = i;
    if (sort_in_bin(box[mi], mi < 3)) {
      task[0] = (task[0] & ~(3u << (2 * mode_step))) | ((mi < 3 ? 1u : 2u) << (2 * mode_step));
      task[1] |= (unsigned)mi << (3 * mode_step);
      mode_step++;
      task[0] = (task[0] & ~(7u << 12)) | ((unsigned)mode_step << 12);
    }
  }
  int code = 0;
  for (int i = 0; i < gc.nb; i++) if (((task[0] >> (2 * i)) & 3) != 1) code |= 1 << (7 - i);
  return code;
}

// ------------------------------------------------------------------------------------------------ env level
template <class C>
D3IL_HD void gen_control_and_physics(const C& c, const GenConsts& gc, EnvState& st, const PushScratch& sc, const double* q_des, const double* qd_des,
                                     double set_width, bool grasp) {
  double tau[NARM], ff[NFING];
  push_control(c, st, q_des, qd_des, set_width, grasp, tau, ff);
  gen_physics_substep(c, gc, st, sc, tau, ff);
}
// Sorting_Env.reset(random=False, context) (sorting.py:545-575): ctx = nb x (pos3, quat4) in the order red_1.., blue_1..
template <class C>
D3IL_HD void gen_env_reset(const C& c, const GenConsts& gc, EnvState& st, const PushScratch& sc, const double* init_qpos, const double* ctx, float* obs) {
  for (int k = 0; k < NARM; k++) { st.q[k] = init_qpos[k]; st.ikq[k] = 0; st.ikqd[k] = 0; }
  st.q[NARM] = 0; st.q[NARM + 1] = 0;
  for (int k = 0; k < NDOF; k++) st.v[k] = 0;
  st.flags = 0; st.step = 0;
  GTASK(0) = 0; GTASK(1) = 0;
  for (int b = 0; b < gc.nb; b++) {
    for (int k = 0; k < 7; k++) GBX(b, k) = ctx[7 * b + k];
    for (int k = 0; k < 6; k++) GBX(b, 7 + k) = 0;
  }
  for (int k = 0; k < 6 * gc.nb + NDOF; k++) GWARM(k) = 0;
  {
    DynOut dyn;
    dynamics(c, st.q, st.v, dyn);
    for (int k = 0; k < NARM; k++) st.bias[k] = dyn.bias[k];
  }
  double zero[NARM] = {0, 0, 0, 0, 0, 0, 0};
  gen_control_and_physics(c, gc, st, sc, init_qpos, zero, 0.001, false);
  double box[6][7];
  sort_collect(gc, sc, box);
  sort_obs_success(gc, box, st.tcp, obs);
}
// before the physics of a step: observation and done (gym_env_wrapper.py:88-90,124-137)
D3IL_HD void sort_step_begin(const GenConsts& gc, EnvState& st, const PushScratch& sc, float* obs, unsigned char* done, int max_steps) {
  double box[6][7];
  sort_collect(gc, sc, box);
  bool succ = sort_obs_success(gc, box, st.tcp, obs);
  bool fin = (st.flags & F_TERMINATED) != 0;
  if (!fin && succ) { st.flags |= F_TERMINATED; fin = true; }
  if (!fin && st.step >= max_steps - 1) fin = true;
  *done = fin ? 1 : 0;
}
// after the physics: success and the completion-order mode code (sorting.py:444-458)
D3IL_HD void sort_step_end(const GenConsts& gc, EnvState& st, const PushScratch& sc, int* mode_code) {
  st.step++;
  double box[6][7]; float dummy[GEN_SORT_OBS];
  sort_collect(gc, sc, box);
  bool succ = sort_obs_success(gc, box, st.tcp, dummy);
  st.flags &= ~F_SUCCESS;
  if (succ) st.flags |= F_SUCCESS | F_TERMINATED;
  unsigned task[2] = {(unsigned)GTASK(0), (unsigned)GTASK(1)};
  *mode_code = sort_check_mode(gc, task, box);
  GTASK(0) = (double)task[0]; GTASK(1) = (double)task[1];
}
template <bool FAST, class C>
D3IL_HD void gen_env_step(const C& c, const GenConsts& gc, EnvState& st, const PushScratch& sc, const double* action, float* obs, unsigned char* done,
                          int* mode_code, int n_substeps, int max_steps) {
  sort_step_begin(gc, st, sc, obs, done, max_steps);
  double des[7];
  make_setpoint(action, des);
  double vwarm[7]; vwarm[6] = 0.0;
#pragma clang loop unroll(disable)
  for (int s = 0; s < n_substeps; s++) {
    D3IL_REFRESH(c, cs);
    ik_update<FAST>(cs, des, des + 3, st.q, st.flags, st.ikq, st.ikqd, vwarm);
    gen_control_and_physics(cs, gc, st, sc, st.ikq, st.ikqd, 0.04, false);
  }
  sort_step_end(gc, st, sc, mode_code);
}

// ------------------------------------------------------------------------------------------------ constants from the blob
D3IL_HOSTFN inline int build_gen_consts(const d3il_model_blob& m, const PandaConsts& pcst, GenConsts& gc, const char** err) {
  std::memset(&gc, 0, sizeof gc);
  gc.nb = m.n_obj;
  if (gc.nb < 1 || gc.nb > GEN_MAXNB) { *err = "unsupported number of task objects"; return -1; }
  auto cube_geom = [&](int body) { for (int g = 0; g < m.ngeom; g++) if (m.geom_body[g] == body && m.geom_contype[g]) return g; return -1; };
  int g0 = cube_geom(m.obj_body[0]);
  if (g0 < 0 || m.geom_type[g0] != D3IL_GEOM_BOX) { *err = "task objects must be boxes"; return -1; }
  for (int k = 0; k < 3; k++) gc.box_half[k] = m.geom_size[g0][k];
  gc.box_mass = m.body_mass[m.obj_body[0]]; gc.box_inertia = m.body_inertia[m.obj_body[0]][0];
  for (int b = 0; b < gc.nb; b++) {
    int bd = m.obj_body[b], g = cube_geom(bd);
    if (g < 0 || m.body_mass[bd] != gc.box_mass || std::fabs(m.body_inertia[bd][1] - gc.box_inertia) > 1e-15 || std::fabs(m.body_inertia[bd][2] - gc.box_inertia) > 1e-15) { *err = "cubes must be identical with isotropic inertia"; return -1; }
    for (int k = 0; k < 3; k++) if (m.geom_size[g][k] != gc.box_half[k] || m.geom_pos[g][k] != 0) { *err = "cubes must be identical and centred"; return -1; }
  }
  gc.box_invw_t = 1.0 / gc.box_mass;
  auto mix = [&](int g1, int g2, int set) {   // mj_contactParam: priority, else solmix average / max friction
    int src = m.geom_priority[g1] > m.geom_priority[g2] ? g1 : (m.geom_priority[g2] > m.geom_priority[g1] ? g2 : -1);
    double sr[2], si[5], fr;
    if (src >= 0) { for (int k = 0; k < 2; k++) sr[k] = m.geom_solref[src][k]; for (int k = 0; k < 5; k++) si[k] = m.geom_solimp[src][k]; fr = m.geom_friction[src][0]; }
    else {
      for (int k = 0; k < 2; k++) sr[k] = 0.5 * (m.geom_solref[g1][k] + m.geom_solref[g2][k]);
      for (int k = 0; k < 5; k++) si[k] = 0.5 * (m.geom_solimp[g1][k] + m.geom_solimp[g2][k]);
      fr = std::fmax(m.geom_friction[g1][0], m.geom_friction[g2][0]);
    }
    double dmax = std::fmin(0.9999, std::fmax(0.0001, si[1])), tc = std::fmax(sr[0], 2 * m.timestep);
    gc.ct_K[set] = 1 / std::fmax(1e-15, dmax * dmax * tc * tc * sr[1] * sr[1]);
    gc.ct_B[set] = 2 / std::fmax(1e-15, dmax * tc);
    for (int k = 0; k < 5; k++) gc.ct_solimp[set][k] = si[k];
    gc.ct_solimp[set][0] = std::fmin(0.9999, std::fmax(0.0001, si[0])); gc.ct_solimp[set][1] = dmax;
    gc.ct_fric[set] = fr;
  };
  // Static boxes a cube can meet: box geoms on joint-less body chains whose collision bits match the cubes' and whose
  // bounding box reaches the modelled workspace - the table_plane footprint shrunk by 8 cm (a cube centre leaving it raises
  // PF_OFF_TABLE), grown by the cube's circumradius, from the table top upwards.  In the Sorting scene: table_plane,
  // support_body, the eight bin walls and the platform; the aluminium profiles around the table edge stay outside.
  const double rcirc = std::sqrt(gc.box_half[0] * gc.box_half[0] + gc.box_half[1] * gc.box_half[1] + gc.box_half[2] * gc.box_half[2]);
  int ns = 0, table = -1;
  double wlo[3] = {0, 0, 0}, whi[3] = {0, 0, 0};
  for (int pass = 0; pass < 2; pass++)
  for (int g = 0; g < m.ngeom; g++) {
    if (m.geom_type[g] != D3IL_GEOM_BOX) continue;
    if (!((m.geom_contype[g] & m.geom_conaffinity[g0]) || (m.geom_contype[g0] & m.geom_conaffinity[g]))) continue;
    bool is_static = true;
    double p[3] = {m.geom_pos[g][0], m.geom_pos[g][1], m.geom_pos[g][2]}, R[9];
    quat2mat(m.geom_quat[g], R);
    for (int bb = m.geom_body[g]; bb > 0; bb = m.body_parent[bb]) {
      if (m.body_jntnum[bb] != 0) { is_static = false; break; }
      double Rb[9], pn[3], Rn[9];
      quat2mat(m.body_quat[bb], Rb);
      for (int i = 0; i < 3; i++) pn[i] = m.body_pos[bb][i] + Rb[3 * i] * p[0] + Rb[3 * i + 1] * p[1] + Rb[3 * i + 2] * p[2];
      for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rn[3 * i + j] = Rb[3 * i] * R[j] + Rb[3 * i + 1] * R[3 + j] + Rb[3 * i + 2] * R[6 + j];
      for (int i = 0; i < 3; i++) p[i] = pn[i];
      for (int i = 0; i < 9; i++) R[i] = Rn[i];
    }
    if (!is_static) continue;
    if (m.geom_margin[g] != 0 || m.geom_gap[g] != 0) { *err = "static boxes with a contact margin are not supported"; return -1; }
    double ext[3];
    for (int i = 0; i < 3; i++) ext[i] = std::fabs(R[3 * i]) * m.geom_size[g][0] + std::fabs(R[3 * i + 1]) * m.geom_size[g][1] + std::fabs(R[3 * i + 2]) * m.geom_size[g][2];
    if (pass == 0) {   // the table top: the 0.49 x 0.98 x 0.001 slab (lab_surrounding.xml:3-4)
      if (std::fabs(m.geom_size[g][0] - 0.49) < 1e-12 && std::fabs(m.geom_size[g][1] - 0.98) < 1e-12 && std::fabs(m.geom_size[g][2] - 0.001) < 1e-12) {
        table = g;
        for (int i = 0; i < 2; i++) { wlo[i] = p[i] - ext[i] + 0.08; whi[i] = p[i] + ext[i] - 0.08; gc.ws_lo[i] = wlo[i]; gc.ws_hi[i] = whi[i]; wlo[i] -= rcirc; whi[i] += rcirc; }
        wlo[2] = p[2] + ext[2] - rcirc;
      }
      continue;
    }
    if (table < 0) { *err = "table slab not found"; return -1; }
    if (p[0] + ext[0] < wlo[0] || p[0] - ext[0] > whi[0] || p[1] + ext[1] < wlo[1] || p[1] - ext[1] > whi[1] || p[2] + ext[2] < wlo[2]) continue;
    if (ns >= GEN_MAXNS) { *err = "too many static boxes"; return -1; }
    for (int k = 0; k < 3; k++) { gc.st_c[ns][k] = p[k]; gc.st_h[ns][k] = m.geom_size[g][k]; }
    for (int k = 0; k < 9; k++) gc.st_R[ns][k] = R[k];
    gc.st_first[ns] = g < g0 ? 1 : 0;
    mix(g, g0, ns);
    ns++;
  }
  gc.ns = ns;
  gc.set_bb = ns; gc.set_rod = ns + 1;
  int g1 = gc.nb > 1 ? cube_geom(m.obj_body[1]) : g0;
  mix(g0, g1, gc.set_bb);
  if (m.rod_geom < 0) { *err = "no rod geom"; return -1; }
  for (int b = 0; b < gc.nb; b++) {   // contact normals follow the model's geom order: cube b before cube b + 1, all cubes before the rod
    int g = cube_geom(m.obj_body[b]);
    if (g > m.rod_geom || (b + 1 < gc.nb && g > cube_geom(m.obj_body[b + 1])) || m.geom_margin[g] != 0) { *err = "unexpected geom order"; return -1; }
  }
  mix(g0, m.rod_geom, gc.set_rod);
  for (int k = 0; k < 3; k++) gc.absent[k] = m.body_pos[m.nbody - 1][k];
  for (int k = 0; k < 4; k++) gc.absent[3 + k] = m.body_quat[m.nbody - 1][k];
  gc.impratio = m.impratio;
  gc.rod_invw = pcst.rod_invweight0;
  return 0;
}

}  // namespace d3il
