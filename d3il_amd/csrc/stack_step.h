// stack_step.h - sub-step of the Stacking task: Panda arm with a USED gripper + three free boxes of individual size + static boxes.
//
// Reference path: CubeStacking_Env.step (stacking.py:331-393) over mujoco.mj_step on the scene of stacking.py:150-156
// (panda_invisible.xml: no rod; finger geoms of class panda:gripper - condim 4, friction 1 / .005 / .0001, margin 1 mm; finger-tip
// boxes with friction 2 / .05 / .0001, solref .01 .5; stacking_objects.py: two 6 cm cubes and a 6 x 10 x 6 cm box, 50 g each).
//
// What is new against the Pushing / Sorting engines: (i) boxes of different size and non-isotropic inertia; (ii) contacts between
// the boxes and geoms of the moving finger bodies - the finger-tip boxes (box-box) and the convex hulls of the finger meshes
// (Minkowski Portal Refinement, one contact per pair like mjc_Convex) - whose Jacobians run over the 7 arm joints and the finger's own
// slide joint; (iii) condim-4 contacts (a torsional row about the normal); (iv) finger <-> finger contacts (an empty gripper closes on
// itself); (v) a joint-space PD law instead of the Cartesian IK controller, so there is no controller wave.
//
// Formulation: one primal Newton problem per sub-step over the 27 dofs [box0 | box1 | box2 | arm 9], as MuJoCo's Newton solver sees
// it: 1/2 (x - a0)' M (x - a0) + sum_i s_i(J_i x - aref_i), elliptic cones, soft-constraint impedances.  M is block diagonal (box:
// mass and body-frame principal inertia; arm: the 9 x 9 matrix of panda_step.h), so the Hessian M + J' Hc J is block sparse:
// blocks are coupled only through contacts, and the Cholesky skips structurally empty blocks (resting boxes cost a 6 x 6 each).
// Contact rows are never stored: a contact record holds position, frame, distance, the two bodies and its parameter set, the rows
// are rebuilt from the record wherever they are needed (a box side: 6 columns from the contact arm; a finger side: z_k x (p - o_k)
// from the joint axes / origins of this sub-step).
//
// Execution: ONE LANE PER ENVIRONMENT (first, correctness-first version): vectors, the packed 27 x 27 Hessian and the kinematic tables
// of a lane's environment sit in LDS (lane-strided), contact records in an HBM scratch area.  Host build: tests/hostcheck.
#pragma once
#include "rigid_common.h"

namespace d3il {

constexpr int SK_NB = 3, SK_NV = 6 * SK_NB + NDOF, SK_ARM0 = 6 * SK_NB, SK_NH = SK_NV * (SK_NV + 1) / 2;   // 27 dofs, 378 packed
constexpr int SK_MAXCON = 32, SK_MAXNS = 4, SK_MAXHV = 72, SK_MAXHANDV = 800;      // SK_MAXHV: capacity for the finger hull (68 vertices) = nine per lane of an eight-lane MPR group
#ifndef D3IL_SK_LANES
#define D3IL_SK_LANES 4
#endif
constexpr int SK_LANES = D3IL_SK_LANES;   // environments per workgroup (one per lane; LDS: 5.6 KiB per environment)
// contact parameter sets
enum { SKS_STATIC = 0 /* + static index */, SKS_BOXBOX = SK_MAXNS, SKS_BOXHULL, SKS_BOXTIP, SKS_HULLHULL, SKS_HULLTIP, SKS_TIPTIP, SKS_BOXHAND, SKS_BOXROD, SKS_N };
// bodies of a contact: boxes 0..2, then
enum { SKB_STATIC = 3, SKB_FINGER = 4 /* + finger: the finger body (hull geom) */, SKB_TIP = 6 /* + finger: the tip body (tip box) */, SKB_HAND = 8 /* the hand body (mesh handv): arm dofs only */, SKB_ROD = 9 /* the rod (Pushing variant): arm dofs only */ };
D3IL_HD int sk_finger_of(int body) { return body >= SKB_HAND ? -1 : (body - SKB_FINGER) & 1; }      // finger whose slide joint moves the body (-1: none)

struct StackSet { double K, B, solimp[5], fric[3], margin; int dim, pad; };
struct StackConsts {
  int nb, ns, hull_nv, pad;
  double box_half[SK_NB][3], box_mass[SK_NB], box_inertia[SK_NB][3];
  double st_c[SK_MAXNS][3], st_h[SK_MAXNS][3], st_R[SK_MAXNS][9];
  // finger geoms in the link-7 frame at finger position 0: the frame moves by f_axis * q_finger
  double tip_R[NFING][9], tip_p[NFING][3], tip_half[3];
  double hull_R[NFING][9], hull_p[NFING][3], hull_center[3], hull_r, tip_r, box_r[SK_NB];     // *_r: bounding radii about the centres
  double hull_v[SK_MAXHV][3];
  double invw_finger[NFING], invw_tip[NFING];     // translational body_invweight0 of the finger / finger-tip bodies
  StackSet set[SKS_N];
  double impratio;
  double target[3], min_dist, grip_thresh;       // stacking_objects.py:17, stacking.py:193, :337
  double ws_lo[2], ws_hi[2];                     // modelled workspace of the box centres (x, y): the table top without its rim
  double hand_R[9], hand_p[3], hand_lo[3], hand_hi[3];   // hand geom (panda_invisible.xml:72, mesh handv) in the link-7 frame; bounding box of its hull in the geom frame: the exact cull of the box <-> hand pairs
  double f_axis0[3];                                     // slide axis of the left finger joint in the link-7 frame
  double hand_center[3], hand_r, invw_hand;              // centroid of the hull (seeds the MPR portal), bounding radius about it, translational body_invweight0 of the hand body
  int hand_nv, hand_pad;
  double hand_v[SK_MAXHANDV][3];                         // convex hull of handv.stl (773 vertices), geom frame
  // rod-robot variants of the engine (panda_rod_invisible.xml, Cartesian controller; today: Aligning): rod cylinder in the link-7 frame, translational
  // body_invweight0 of the rod body
  int variant, var_pad;
  double rod_c7[3], rod_u7[3], rod_r, rod_h, invw_rod;
  double box_invw[SK_NB];      // translational body_invweight0 of the free bodies (1 / mass for a body whose centre of mass is its origin)
  // Aligning variant (variant 2: the rod robot and ONE free compound body = block 0, robot_push_box.xml: a plate carrying four walls - AL_NG box
  // geoms on one body, centre of mass al_c in the body frame).  The engine works in CENTRE-OF-MASS coordinates of that body (mass matrix
  // diagonal, gravity torque-free): stack_pre_kin converts MuJoCo's (origin position, origin velocity) to them, stack_substep_post converts the
  // accelerations back before the Euler step, and the contact rows' reference accelerations carry the centripetal term that separates
  // J qacc of the two coordinate systems.
  int al_ng, al_pad;
  double al_gpos[5][3], al_ghalf[5][3], al_gr[5];      // geom centres in the body frame, half sizes, bounding radii
  double al_c[3], al_r;                                // centre of mass in the body frame; bounding radius of the whole body about its origin
  int al_set_static[5], al_set_rod[5];                 // contact parameter set of geom g against static s (+ s) / against the rod
};
enum { SKV_STACKING = 0, SKV_ALIGNING = 2 };      // (1 was the Pushing variant of rounds 3 - 5; Pushing runs on the generic engine)
constexpr int AL_NG = 5;
// rod-robot variants: the finger-geom tables of the t area are not needed; their place holds the rod pose and the controller state
constexpr int SV_ROD = 0 /* + ST_TIPR: rod centre[3], axis[3] */, SV_IKQ = 6, SV_IKQD = 13, SV_DES = 20 /* desired pose pos[3] quat[4] */, SV_VWARM = 27 /* 7 */, SV_END = 34;
// Aligning variant: + the centripetal acceleration of the body origin relative to the centre of mass, w x (w x R c) in the world frame [3], and the hold flag of a reset
constexpr int SV_CEN = 34, SV_HOLD = 37, SV_END2 = 38;

// flag bits of the Stacking task (EnvState::flags).  F_TERMINATED / F_SUCCESS / F_SOLVER_FAIL keep their positions.
enum : unsigned {
  SKF_NMODE_MASK = 0x3u,          // number of boxes that have reached the target so far (stacking.py:395-419)
  SKF_IND_SHIFT = 2,              // min_inds[3], two bits each
  SKF_WARM_VALID = 1u << 8,       // the warm-start rows hold the accelerations of the previous sub-step
  SKF_CON_OVERFLOW = 1u << 18,    // more than SK_MAXCON contacts in one sub-step (extra contacts dropped)
  SKF_OFF_TABLE = 1u << 19,       // a box left the modelled part of the table
  SKF_HAND_NEAR = 1u << 20,       // a box reached the hand mesh (a pair this engine does not evaluate)
};
// arm q[9] v[9] bias[7] tcp[3] | boxes (pos3 quat4 vel6) x 3 | the solver's warm start qacc[27] (MuJoCo: qacc_warmstart)
constexpr int SK_STATE_BOX = 28, SK_STATE_WARM = SK_STATE_BOX + 13 * SK_NB, SK_STATE_F64 = SK_STATE_WARM + SK_NV;
constexpr int SK_OBS = 12, SK_ACT = 8;

#if defined(__HIPCC__)
__constant__ StackConsts g_stack_consts;
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define D3IL_STACK_CONSTS(in, name) const StackConsts& name = g_stack_consts; (void)in
#else
#define D3IL_STACK_CONSTS(in, name) const StackConsts& name = in
#endif

// scratch views of one environment: t area (LDS on the device) and g area (HBM)
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(3))) double sk_lds_double;
typedef __attribute__((address_space(1))) double sk_glb_double;
#else
typedef double sk_lds_double;
typedef double sk_glb_double;
#endif
// both areas are contiguous per environment (t: ST_SIZE doubles, odd => the lanes of a wave hit different LDS banks; g: SG_SIZE doubles),
// so that the wave-cooperative solver (sk_solve_coop) can address one environment's data from all lanes
struct StackScratch { sk_lds_double* t; sk_glb_double* g; };
#define SL(i) sc.t[(i)]
#define SG(i) sc.g[(i)]
// t area
constexpr int ST_H = 0, ST_G = SK_NH, ST_P = ST_G + SK_NV;      // solver work vectors: packed Hessian, gradient, direction ("head")
constexpr int ST_HEAD = ST_P + SK_NV;                             // 432
constexpr int ST_X = ST_HEAD, ST_A0 = ST_X + SK_NV, ST_VEL = ST_A0 + SK_NV;
constexpr int ST_M = ST_VEL + SK_NV;            // arm mass matrix, packed lower 45
constexpr int ST_BR = ST_M + 45;                // box rotation matrices 3 x 9
constexpr int ST_BP = ST_BR + 27;               // box positions 3 x 3
constexpr int ST_Z = ST_BP + 9;                 // world joint axes 7 x 3
constexpr int ST_O = ST_Z + 21;                 // world joint origins 7 x 3
constexpr int ST_FAX = ST_O + 21;               // world finger slide axes 2 x 3
constexpr int ST_TIPR = ST_FAX + 6, ST_TIPP = ST_TIPR + 18, ST_HULR = ST_TIPP + 6, ST_HULP = ST_HULR + 18;
constexpr int ST_LIM = ST_HULP + 6;             // per arm dof: sign, D, aref
constexpr int ST_AUX = ST_LIM + 27;             // finger opening q7 + q8 (gate of the finger <-> finger pairs), spare
constexpr int ST_SIZE = ST_AUX + 2;             // 719 (odd)
// g area: contact records
constexpr int SREC = 36;    // pos[3] frame[9] dist bodyA bodyB set | aref[4] D[4] mu | jar[4] jp[4] | pad
constexpr int SG_DIAG = SK_MAXCON * SREC;   // diagnostics of the last sub-step: Newton iterations, final max |gradient|, converged, contacts
constexpr int SG_SIZE = SG_DIAG + 24;      // [4 .. 11]: clock ticks per phase, accumulated (diagnostics build -DD3IL_DEVICE_STATS only)

#if defined(D3IL_DEVICE_STATS) && defined(__HIP_DEVICE_COMPILE__)
#define SK_TIC unsigned long long sk_t0_ = wall_clock64()
#define SK_TOC(slot) do { unsigned long long t_ = wall_clock64(); SG(SG_DIAG + 4 + (slot)) += (double)(t_ - sk_t0_); sk_t0_ = t_; } while (0)
#else
#define SK_TIC ((void)0)
#define SK_TOC(slot) ((void)0)
#endif
// ------------------------------------------------------------------------------------------------ convex pairs: MPR
// Same algorithm as the oracle's mpr_penetration (libccd's ccdMPRPenetration as MuJoCo 2.3.2 runs it for mesh geoms [ext]); the
// tie rule of the support functions (lowest index within 1e-10, box components >= -1e-10 positive) makes the portal independent
// of round-off in flat-on-flat configurations.  Written with exact divisions / square roots: the portal logic branches on signs.
struct SkShape { const double* R; const double* p; const double* half; int hull; };   // hull: 0 a box, 1 the finger hull, 2 the hand hull
struct SkPt { double v[3], v1[3], v2[3]; };
D3IL_HD void sk_support1(const StackConsts& kc_, const SkShape& s, const double* dir, double margin, double* out) {
  D3IL_STACK_CONSTS(kc_, kc);
  const double* R = s.R;
  double dl[3] = {R[0] * dir[0] + R[3] * dir[1] + R[6] * dir[2], R[1] * dir[0] + R[4] * dir[1] + R[7] * dir[2], R[2] * dir[0] + R[5] * dir[1] + R[8] * dir[2]};
  double loc[3];
  if (s.hull) {
    // lowest-index vertex within 1e-10 of the maximum: pass 1 finds the maximum and the first vertex attaining it, pass 2 looks for an
    // earlier vertex inside the tolerance.  Four vertices per iteration: the (wave-uniform) vertex table comes through the scalar
    // cache, one wait per four vertices instead of one per vertex.  hull: 1 the finger hull, 2 the hand hull.
    const int nv = s.hull == 2 ? kc.hand_nv : kc.hull_nv;
    const double (*tab)[3] = s.hull == 2 ? kc.hand_v : kc.hull_v;
    auto dotv = [&](int i) { return tab[i][0] * dl[0] + tab[i][1] * dl[1] + tab[i][2] * dl[2]; };
    double bd = -1e300;
    int imax = 0, i = 0;
    for (; i + 4 <= nv; i += 4) {
      const double d0 = dotv(i), d1 = dotv(i + 1), d2 = dotv(i + 2), d3 = dotv(i + 3);
      if (d0 > bd) { bd = d0; imax = i; }
      if (d1 > bd) { bd = d1; imax = i + 1; }
      if (d2 > bd) { bd = d2; imax = i + 2; }
      if (d3 > bd) { bd = d3; imax = i + 3; }
    }
    for (; i < nv; i++) { const double d = dotv(i); if (d > bd) { bd = d; imax = i; } }
    int best = imax;
    const double thr = bd - 1e-10;
    for (i = 0; i + 4 <= imax; i += 4) {
      const double d0 = dotv(i), d1 = dotv(i + 1), d2 = dotv(i + 2), d3 = dotv(i + 3);
      const int hit = d0 >= thr ? i : (d1 >= thr ? i + 1 : (d2 >= thr ? i + 2 : (d3 >= thr ? i + 3 : -1)));
      if (hit >= 0) { best = hit; break; }
    }
    if (best == imax) for (; i < imax; i++) if (dotv(i) >= thr) { best = i; break; }
    loc[0] = tab[best][0]; loc[1] = tab[best][1]; loc[2] = tab[best][2];
  } else {
#pragma unroll
    for (int k = 0; k < 3; k++) loc[k] = dl[k] >= -1e-10 ? s.half[k] : -s.half[k];
  }
#pragma unroll
  for (int k = 0; k < 3; k++) out[k] = R[3 * k] * loc[0] + R[3 * k + 1] * loc[1] + R[3 * k + 2] * loc[2] + s.p[k] + 0.5 * margin * dir[k];
}
D3IL_HD void sk_support(const StackConsts& kc, const SkShape& a, const SkShape& b, const double* dir, double margin, SkPt& pt) {
  double nd[3] = {-dir[0], -dir[1], -dir[2]};
  sk_support1(kc, a, dir, margin, pt.v1); sk_support1(kc, b, nd, margin, pt.v2);
#pragma unroll
  for (int k = 0; k < 3; k++) pt.v[k] = pt.v1[k] - pt.v2[k];
}
D3IL_HD bool sk_zero(double x) { return fabs(x) < 2.220446049250313e-16; }
D3IL_HD void sk_norm3(double* a) { double n = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); a[0] /= n; a[1] /= n; a[2] /= n; }
D3IL_HD void sk_portal_dir(const SkPt* P, double* dir) {
  double a[3] = {P[2].v[0] - P[1].v[0], P[2].v[1] - P[1].v[1], P[2].v[2] - P[1].v[2]}, b[3] = {P[3].v[0] - P[1].v[0], P[3].v[1] - P[1].v[1], P[3].v[2] - P[1].v[2]};
  cross3(a, b, dir); sk_norm3(dir);
}
// dst = c ? src : dst, element by element: value selects keep the portal in registers (a conditional struct store is merged by the
// compiler into one store through a SELECTED POINTER, which pins the whole portal array in private memory)
D3IL_HD void sk_sel(SkPt& dst, const SkPt& src, bool c) {
#pragma unroll
  for (int k = 0; k < 3; k++) { dst.v[k] = c ? src.v[k] : dst.v[k]; dst.v1[k] = c ? src.v1[k] : dst.v1[k]; dst.v2[k] = c ? src.v2[k] : dst.v2[k]; }
}
D3IL_HD void sk_expand(SkPt* P, const SkPt& v4) {
  double w[3]; cross3(v4.v, P[0].v, w);
  const bool b1 = dot3(P[1].v, w) > 0, b2 = dot3(P[2].v, w) > 0, b3 = dot3(P[3].v, w) > 0;
  // b1 ? (b2 ? P[1] : P[3]) : (b3 ? P[2] : P[1]) = v4
  sk_sel(P[1], v4, (b1 && b2) || (!b1 && !b3));
  sk_sel(P[2], v4, !b1 && b3);
  sk_sel(P[3], v4, b1 && !b2);
}
D3IL_HD bool sk_reach_tol(const SkPt* P, const SkPt& v4, const double* dir) {
  double dv4 = dot3(v4.v, dir);
  double d = fmin(dv4 - dot3(P[1].v, dir), fmin(dv4 - dot3(P[2].v, dir), dv4 - dot3(P[3].v, dir)));
  return d < 1e-6 || sk_zero(d - 1e-6);
}
D3IL_HD void sk_tri_closest_origin(const double* a, const double* b, const double* c, double* out) {   // Ericson, RTCD 5.1.5
  double ab[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, ac[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]}, ap[3] = {-a[0], -a[1], -a[2]};
  double d1 = dot3(ab, ap), d2 = dot3(ac, ap);
  if (d1 <= 0 && d2 <= 0) { out[0] = a[0]; out[1] = a[1]; out[2] = a[2]; return; }
  double bp[3] = {-b[0], -b[1], -b[2]}, d3 = dot3(ab, bp), d4 = dot3(ac, bp);
  if (d3 >= 0 && d4 <= d3) { out[0] = b[0]; out[1] = b[1]; out[2] = b[2]; return; }
  double vc = d1 * d4 - d3 * d2;
  if (vc <= 0 && d1 >= 0 && d3 <= 0) { double v = d1 / (d1 - d3); for (int k = 0; k < 3; k++) out[k] = a[k] + v * ab[k]; return; }
  double cp[3] = {-c[0], -c[1], -c[2]}, d5 = dot3(ab, cp), d6 = dot3(ac, cp);
  if (d6 >= 0 && d5 <= d6) { out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; return; }
  double vb = d5 * d2 - d1 * d6;
  if (vb <= 0 && d2 >= 0 && d6 <= 0) { double w = d2 / (d2 - d6); for (int k = 0; k < 3; k++) out[k] = a[k] + w * ac[k]; return; }
  double va = d3 * d6 - d5 * d4;
  if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) { double w = (d4 - d3) / ((d4 - d3) + (d5 - d6)); for (int k = 0; k < 3; k++) out[k] = b[k] + w * (c[k] - b[k]); return; }
  double den = 1 / (va + vb + vc), v = vb * den, w = vc * den;
  for (int k = 0; k < 3; k++) out[k] = a[k] + ab[k] * v + ac[k] * w;
}
// out = {dist, pos[3], normal[3]} (normal from shape a to shape b); false when the inflated shapes do not overlap
#if defined(SK_MPR_STATS)
#define SK_MPR_COUNT(x) (x)
#else
#define SK_MPR_COUNT(x) ((void)0)
#endif
// SUP(dir, pt): support point of the Minkowski difference a - b in direction dir (sk_support on one lane; the device collision phase
// spreads the hull vertices of a pair over a group of eight lanes)
template <class SHA, class SHB, class SUP>
D3IL_HD bool sk_mpr_t(const StackConsts& kc_, const SHA a, const SHB b, double margin, double* out, SUP sup) {
  D3IL_STACK_CONSTS(kc_, kc);
  SK_MPR_COUNT(g_calls++);
  SkPt P[4], v4;
  double dir[3], va[3], vb[3];
  {   // geom centres (written out per shape: a pointer selected at run time into P[] would pin the whole portal in private memory)
    const double* ca = a.hull == 2 ? kc.hand_center : kc.hull_center;
    const double* cb = b.hull == 2 ? kc.hand_center : kc.hull_center;
    if (a.hull) { for (int k = 0; k < 3; k++) P[0].v1[k] = a.R[3 * k] * ca[0] + a.R[3 * k + 1] * ca[1] + a.R[3 * k + 2] * ca[2] + a.p[k]; }
    else { P[0].v1[0] = a.p[0]; P[0].v1[1] = a.p[1]; P[0].v1[2] = a.p[2]; }
    if (b.hull) { for (int k = 0; k < 3; k++) P[0].v2[k] = b.R[3 * k] * cb[0] + b.R[3 * k + 1] * cb[1] + b.R[3 * k + 2] * cb[2] + b.p[k]; }
    else { P[0].v2[0] = b.p[0]; P[0].v2[1] = b.p[1]; P[0].v2[2] = b.p[2]; }
  }
  for (int k = 0; k < 3; k++) P[0].v[k] = P[0].v1[k] - P[0].v2[k];
  if (sk_zero(P[0].v[0]) && sk_zero(P[0].v[1]) && sk_zero(P[0].v[2])) P[0].v[0] += 10 * 2.220446049250313e-16;
  for (int k = 0; k < 3; k++) dir[k] = -P[0].v[k];
  sk_norm3(dir);
  sup(dir, P[1]);
  double dot = dot3(P[1].v, dir);
  if (sk_zero(dot) || dot < 0) return false;
  cross3(P[0].v, P[1].v, dir);
  if (sk_zero(dot3(dir, dir))) {
    if (sk_zero(P[1].v[0]) && sk_zero(P[1].v[1]) && sk_zero(P[1].v[2])) return false;
    double depth = sqrt(dot3(P[1].v, P[1].v));
    for (int k = 0; k < 3; k++) { out[1 + k] = 0.5 * (P[1].v1[k] + P[1].v2[k]); out[4 + k] = P[1].v[k] / depth; }
    out[0] = margin - depth;
    return true;
  }
  sk_norm3(dir);
  sup(dir, P[2]);
  dot = dot3(P[2].v, dir);
  if (sk_zero(dot) || dot < 0) return false;
  for (int k = 0; k < 3; k++) { va[k] = P[1].v[k] - P[0].v[k]; vb[k] = P[2].v[k] - P[0].v[k]; }
  cross3(va, vb, dir); sk_norm3(dir);
  {
    const bool sw = dot3(dir, P[0].v) > 0;
    const SkPt t1 = P[1], t2 = P[2];
    sk_sel(P[1], t2, sw); sk_sel(P[2], t1, sw);
    if (sw) { dir[0] = -dir[0]; dir[1] = -dir[1]; dir[2] = -dir[2]; }
  }
  for (int guard = 0; guard < 100; guard++) {
    SK_MPR_COUNT(g_disc++);
    sup(dir, P[3]);
    dot = dot3(P[3].v, dir);
    if (sk_zero(dot) || dot < 0) return false;
    cross3(P[1].v, P[3].v, va); dot = dot3(va, P[0].v);
    const bool c2 = dot < 0 && !sk_zero(dot);
    cross3(P[3].v, P[2].v, va); dot = dot3(va, P[0].v);      // evaluated with the portal BEFORE the replacement; only used when c2 is false
    const bool c1 = !c2 && dot < 0 && !sk_zero(dot);
    { const SkPt t3 = P[3]; sk_sel(P[2], t3, c2); sk_sel(P[1], t3, c1); }
    if (!c2 && !c1) break;
    for (int k = 0; k < 3; k++) { va[k] = P[1].v[k] - P[0].v[k]; vb[k] = P[2].v[k] - P[0].v[k]; }
    cross3(va, vb, dir); sk_norm3(dir);
  }
  for (int guard = 0; ; guard++) {          // refine the portal until it encloses the origin
    sk_portal_dir(P, dir);
    dot = dot3(dir, P[1].v);
    if (sk_zero(dot) || dot > 0) break;
    SK_MPR_COUNT(g_ref++);
    sup(dir, v4);
    dot = dot3(v4.v, dir);
    if (!(sk_zero(dot) || dot > 0) || sk_reach_tol(P, v4, dir) || guard > 100) return false;
    sk_expand(P, v4);
  }
  for (int it = 0; ; it++) {                // penetration
    sk_portal_dir(P, dir);
    SK_MPR_COUNT((g_pen++, g_maxpen = it + 1 > g_maxpen ? it + 1 : g_maxpen));
    sup(dir, v4);
    if (sk_reach_tol(P, v4, dir) || it > 50) {
      double w[3]; sk_tri_closest_origin(P[1].v, P[2].v, P[3].v, w);
      double depth = sqrt(dot3(w, w));
      if (sk_zero(w[0]) && sk_zero(w[1]) && sk_zero(w[2])) { w[0] = dir[0]; w[1] = dir[1]; w[2] = dir[2]; }
      sk_norm3(w);
      double bb[4], t[3], sum;
      cross3(P[1].v, P[2].v, t); bb[0] = dot3(t, P[3].v);
      cross3(P[3].v, P[2].v, t); bb[1] = dot3(t, P[0].v);
      cross3(P[0].v, P[1].v, t); bb[2] = dot3(t, P[3].v);
      cross3(P[2].v, P[1].v, t); bb[3] = dot3(t, P[0].v);
      sum = bb[0] + bb[1] + bb[2] + bb[3];
      if (sk_zero(sum) || sum < 0) {
        bb[0] = 0;
        cross3(P[2].v, P[3].v, t); bb[1] = dot3(t, dir);
        cross3(P[3].v, P[1].v, t); bb[2] = dot3(t, dir);
        cross3(P[1].v, P[2].v, t); bb[3] = dot3(t, dir);
        sum = bb[1] + bb[2] + bb[3];
      }
      double inv = 1 / sum;
      for (int k = 0; k < 3; k++) {
        double p1 = bb[0] * P[0].v1[k] + bb[1] * P[1].v1[k] + bb[2] * P[2].v1[k] + bb[3] * P[3].v1[k];
        double p2 = bb[0] * P[0].v2[k] + bb[1] * P[1].v2[k] + bb[2] * P[2].v2[k] + bb[3] * P[3].v2[k];
        out[1 + k] = 0.5 * (p1 + p2) * inv; out[4 + k] = w[k];
      }
      out[0] = margin - depth;
      return true;
    }
    sk_expand(P, v4);
  }
}
D3IL_NOINLINE inline bool sk_mpr(const StackConsts& kc_, const SkShape a, const SkShape b, double margin, double* out) {
  return sk_mpr_t(kc_, a, b, margin, out, [&](const double* dir, SkPt& pt) { sk_support(kc_, a, b, dir, margin, pt); });
}

// ------------------------------------------------------------------------------------------------ contact rows
// elliptic cone of dimension dim (3 or 4): force and Hessian block at the row residuals jar; D[r] = 1 / R[r], fr[j] = friction
// coefficient of row j + 1 (tangent, tangent, torsional).  Zones as in MuJoCo's PGS / Newton cone [ext]; mirrors cone_eval (dim 3).
// imu = 1 / max(1e-15, mu^2 (1 + mu^2)) is a constant of the contact
D3IL_HD void sk_cone_pre(int dim, const double* jar, const double* D, double mu, double imu, const double* fr, double* force, double* Hc /* 4 x 4 */) {
  // all loops run to the fixed bound 4 with a row predicate: every index is a compile-time constant, nothing is addressed in private memory
#pragma unroll
  for (int i = 0; i < 16; i++) Hc[i] = 0;
  double U[4] = {jar[0] * mu, 0, 0, 0}, T2 = 0;
#pragma unroll
  for (int j = 1; j < 4; j++) if (j < dim) { U[j] = jar[j] * fr[j - 1]; T2 += U[j] * U[j]; }
  const double N = U[0], iT = T2 > 0 ? rsqrtd(T2) : 0.0, T = T2 * iT;      // 1 / T by v_rsq_f64 + Newton steps: no square root, no divisions below
  if (N >= mu * T || (T <= 0 && N >= 0)) { force[0] = force[1] = force[2] = force[3] = 0; return; }
  if (mu * N + T <= 0 || (T <= 0 && N < 0)) {
#pragma unroll
    for (int j = 0; j < 4; j++) { force[j] = j < dim ? -D[j] * jar[j] : 0.0; if (j < dim) Hc[5 * j] = D[j]; }
    return;
  }
  const double Dm = D[0] * imu, NmT = N - mu * T;
  double g[4] = {mu, 0, 0, 0}, Un[4] = {0, 0, 0, 0};      // Un = U / T
#pragma unroll
  for (int j = 1; j < 4; j++) if (j < dim) { Un[j] = U[j] * iT; g[j] = -mu * fr[j - 1] * Un[j]; }
#pragma unroll
  for (int j = 0; j < 4; j++) force[j] = j < dim ? -Dm * NmT * g[j] : 0.0;
  const double kT = NmT * (-mu) * iT;
#pragma unroll
  for (int a = 0; a < 4; a++) {
#pragma unroll
    for (int b = 0; b < 4; b++) {
      if (a >= dim || b >= dim) continue;
      double h = g[a] * g[b];
      if (a > 0 && b > 0) h += kT * fr[a - 1] * fr[b - 1] * ((a == b ? 1.0 : 0.0) - Un[a] * Un[b]);
      Hc[4 * a + b] = Dm * h;
    }
  }
}
D3IL_HD void sk_cone(int dim, const double* jar, const double* D, double mu, const double* fr, double* force, double* Hc /* 4 x 4 */) {
  sk_cone_pre(dim, jar, D, mu, 1.0 / fmax(1e-15, mu * mu * (1 + mu * mu)), fr, force, Hc);
}
// Rows of a contact.  The four body pairings have compile-time sizes, so their row blocks live in registers:
//   (NA, NB) = (0, 6) static-box, (6, 6) box-box, (6, 9) box-finger, (0, 9) finger-finger (one row set J(finger B) - J(finger A)).
// A = body 1 (enters with -), B = body 2 (+); a box block has 6 columns, the arm block 9.  Blocks are ordered box 0 | 1 | 2 | arm and
// body 1 precedes body 2 in the model's geom order, so the cross block of the Hessian is always H[B rows][A columns].
template <int NA, int NB> struct SkRows { int oa, ob, dim; double A[4][NA ? NA : 1], B[4][NB]; };
D3IL_HD void sk_box_rows(const StackScratch sc, int body, const double* pos, const double* frame, int dim, double (*J)[6]) {
  double R[9], r[3];
#pragma unroll
  for (int k = 0; k < 9; k++) R[k] = SL(ST_BR + 9 * body + k);
#pragma unroll
  for (int k = 0; k < 3; k++) r[k] = pos[k] - SL(ST_BP + 3 * body + k);
#pragma unroll
  for (int rr = 0; rr < 3; rr++) box_row_r(R, r, frame + 3 * rr, J[rr]);
  J[3][0] = J[3][1] = J[3][2] = 0;      // torsional row (dim 4): relative angular velocity about the normal; box angular dofs are body axes
  J[3][3] = R[0] * frame[0] + R[3] * frame[1] + R[6] * frame[2];
  J[3][4] = R[1] * frame[0] + R[4] * frame[1] + R[7] * frame[2];
  J[3][5] = R[2] * frame[0] + R[5] * frame[1] + R[8] * frame[2];
  (void)dim;
}
D3IL_HD void sk_arm_rows(const StackScratch sc, int f, double sign, const double* pos, const double* frame, double (*J)[NDOF], bool accumulate) {
#pragma unroll
  for (int k = 0; k < NARM; k++) {
    double z[3] = {SL(ST_Z + 3 * k), SL(ST_Z + 3 * k + 1), SL(ST_Z + 3 * k + 2)};
    double d[3] = {pos[0] - SL(ST_O + 3 * k), pos[1] - SL(ST_O + 3 * k + 1), pos[2] - SL(ST_O + 3 * k + 2)}, col[3];
    cross3(z, d, col);
#pragma unroll
    for (int rr = 0; rr < 3; rr++) { const double v = sign * dot3(frame + 3 * rr, col); J[rr][k] = accumulate ? J[rr][k] + v : v; }
    const double w = sign * dot3(frame, z);
    J[3][k] = accumulate ? J[3][k] + w : w;
  }
#pragma unroll
  for (int g = 0; g < NFING; g++) {
    double ax[3] = {SL(ST_FAX + 3 * g), SL(ST_FAX + 3 * g + 1), SL(ST_FAX + 3 * g + 2)};
#pragma unroll
    for (int rr = 0; rr < 3; rr++) { const double v = g == f ? sign * dot3(frame + 3 * rr, ax) : 0.0; J[rr][NARM + g] = accumulate ? J[rr][NARM + g] + v : v; }
    if (!accumulate) J[3][NARM + g] = 0;
  }
}
template <int NA, int NB>
D3IL_HD void sk_build_rows_rec(const StackConsts& kc_, const StackScratch sc, const double* rec /* pos3 frame9 dist a b set */, SkRows<NA, NB>& R, int* set_out) {
  D3IL_STACK_CONSTS(kc_, kc);
  const int a = (int)rec[13], b = (int)rec[14], set = (int)rec[15];
  *set_out = set; R.dim = kc.set[set].dim;
  if constexpr (NA == 6) { R.oa = 6 * a; sk_box_rows(sc, a, rec, rec + 3, R.dim, R.A); } else R.oa = 0;
  if constexpr (NB == 6) { R.ob = 6 * b; sk_box_rows(sc, b, rec, rec + 3, R.dim, R.B); }
  else {
    R.ob = SK_ARM0;
    sk_arm_rows(sc, sk_finger_of(b), 1.0, rec, rec + 3, R.B, false);
    if constexpr (NA == 0) sk_arm_rows(sc, sk_finger_of(a), -1.0, rec, rec + 3, R.B, true);     // finger <-> finger
  }
}
template <int NA, int NB>
D3IL_HD void sk_build_rows(const StackConsts& kc_, const StackScratch sc, int ci, SkRows<NA, NB>& R, int* set_out) {
  const int base = ci * SREC;
  double rec[16];
#pragma unroll
  for (int k = 0; k < 16; k++) rec[k] = SG(base + k);
  sk_build_rows_rec(kc_, sc, rec, R, set_out);
}
template <int NA, int NB> D3IL_HD double sk_dot(const StackScratch sc, const SkRows<NA, NB>& R, int r, int vec) {
  double s = 0;
#pragma unroll
  for (int k = 0; k < NB; k++) s += R.B[r][k] * SL(vec + R.ob + k);
  if constexpr (NA > 0) {
#pragma unroll
    for (int k = 0; k < NA; k++) s -= R.A[r][k] * SL(vec + R.oa + k);
  }
  return s;
}
// friction coefficients of the rows 1 .. 3 of a contact: tangent, tangent, torsional (mjContact.friction[0, 1, 2] of MuJoCo's
// 5-vector (slide, slide, spin, roll, roll))
D3IL_HD void sk_row_fric(const StackSet& ps, double* fr) { fr[0] = ps.fric[0]; fr[1] = ps.fric[0]; fr[2] = ps.fric[1]; }
// accumulation into the t area (device: LDS atomic add, fire and forget - the lane owns its column, so there is no contention)
#if defined(__HIP_DEVICE_COMPILE__)
#define SL_ADD(i, v) ((void)__hip_atomic_fetch_add(&SL(i), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT))
#else
#define SL_ADD(i, v) (SL(i) += (v))
#endif
D3IL_HD int sk_kind(int a, int b) { return b < SK_NB ? (a == SKB_STATIC ? 0 : 1) : (a < SK_NB ? 2 : 3); }
D3IL_HD int sk_blk_of(int body) { return body == SKB_STATIC ? -1 : (body < SK_NB ? body : SK_NB); }
D3IL_HD int sk_blk(int dof) { return dof >= SK_ARM0 ? SK_NB : dof / 6; }
D3IL_HD int sk_blk0(int b) { return b < SK_NB ? 6 * b : SK_ARM0; }
D3IL_HD int sk_blkn(int b) { return b < SK_NB ? 6 : NDOF; }
D3IL_HD double sk_Mv(const StackConsts& kc_, const StackScratch sc, int i, int va, int vb) {   // (M (v_a - v_b))_i of the block-diagonal mass matrix
  D3IL_STACK_CONSTS(kc_, kc);
  if (i < SK_ARM0) {
    const int b = i / 6, k = i % 6;
    return (k < 3 ? kc.box_mass[b] : kc.box_inertia[b][k - 3]) * (SL(va + i) - SL(vb + i));
  }
  const int a = i - SK_ARM0;
  double s = 0;
  for (int k = 0; k < NDOF; k++) s += SL(ST_M + (a >= k ? tri(a, k) : tri(k, a))) * (SL(va + SK_ARM0 + k) - SL(vb + SK_ARM0 + k));
  return s;
}

// gradient / Hessian contribution of one contact at x (ST_X): g -= J' f, H += J' Hc J; the row residuals are cached in the record
template <int NA, int NB>
D3IL_NOINLINE inline void sk_contact_gh(const StackConsts& kc_, const StackScratch sc, int ci, bool with_h) {
  D3IL_STACK_CONSTS(kc_, kc);
  SkRows<NA, NB> R; int set;
  sk_build_rows(kc, sc, ci, R, &set);
  const int base = ci * SREC, dim = R.dim;
  double jar[4] = {0, 0, 0, 0}, D[4], f[4], Hc[16], fr[3];
  for (int r = 0; r < 4; r++) { if (r < dim) jar[r] = sk_dot(sc, R, r, ST_X) - SG(base + 16 + r); D[r] = SG(base + 20 + r); SG(base + 25 + r) = jar[r]; }
  sk_row_fric(kc.set[set], fr);
  sk_cone(dim, jar, D, SG(base + 24), fr, f, Hc);
  bool any = false;
#pragma unroll
  for (int i = 0; i < 16; i++) any = any || Hc[i] != 0;
  if (!any) return;                      // contact in the top zone of its cone: no force, no curvature
#pragma unroll
  for (int k = 0; k < NB; k++) { double acc = 0; for (int r = 0; r < 4; r++) acc += R.B[r][k] * f[r]; SL_ADD(ST_G + R.ob + k, -acc); }
  if constexpr (NA > 0) {
#pragma unroll
    for (int k = 0; k < NA; k++) { double acc = 0; for (int r = 0; r < 4; r++) acc += R.A[r][k] * f[r]; SL_ADD(ST_G + R.oa + k, acc); }
  }
  if (!with_h) return;                   // gradient-only pass (warm-start acceptance test)
  // H blocks (rows beyond dim carry zero force / curvature: f and Hc are zero there)
#pragma unroll
  for (int i = 0; i < NB; i++) {
    double t[4];
#pragma unroll
    for (int q = 0; q < 4; q++) t[q] = R.B[0][i] * Hc[q] + R.B[1][i] * Hc[4 + q] + R.B[2][i] * Hc[8 + q] + R.B[3][i] * Hc[12 + q];
#pragma unroll
    for (int k = 0; k <= i; k++) SL_ADD(ST_H + tri(R.ob + i, R.ob + k), t[0] * R.B[0][k] + t[1] * R.B[1][k] + t[2] * R.B[2][k] + t[3] * R.B[3][k]);
    if constexpr (NA > 0) {
#pragma unroll
      for (int k = 0; k < NA; k++) SL_ADD(ST_H + tri(R.ob + i, R.oa + k), -(t[0] * R.A[0][k] + t[1] * R.A[1][k] + t[2] * R.A[2][k] + t[3] * R.A[3][k]));
    }
  }
  if constexpr (NA > 0) {
#pragma unroll
    for (int i = 0; i < NA; i++) {
      double t[4];
#pragma unroll
      for (int q = 0; q < 4; q++) t[q] = R.A[0][i] * Hc[q] + R.A[1][i] * Hc[4 + q] + R.A[2][i] * Hc[8 + q] + R.A[3][i] * Hc[12 + q];
#pragma unroll
      for (int k = 0; k <= i; k++) SL_ADD(ST_H + tri(R.oa + i, R.oa + k), t[0] * R.A[0][k] + t[1] * R.A[1][k] + t[2] * R.A[2][k] + t[3] * R.A[3][k]);
    }
  }
}
// J p of one contact for the line search (cached in the record), or J v -> reference acceleration and regularisation (mode 1)
template <int NA, int NB>
D3IL_NOINLINE inline void sk_contact_dot(const StackConsts& kc_, const StackScratch sc, int ci, int mode) {
  D3IL_STACK_CONSTS(kc_, kc);
  SkRows<NA, NB> R; int set;
  sk_build_rows(kc, sc, ci, R, &set);
  const int base = ci * SREC;
  if (mode == 0) {
    for (int r = 0; r < 4; r++) SG(base + 29 + r) = r < R.dim ? sk_dot(sc, R, r, ST_P) : 0.0;
    return;
  }
  // mj_makeImpedance for an elliptic contact [ext]
  const StackSet& ps = kc.set[set];
  const double dist = SG(base + 12);
  const double imp = impedance(ps.solimp, dist - ps.margin);
  const int a = (int)SG(base + 13), b = (int)SG(base + 14);
  auto invw = [&](int body) { return body == SKB_STATIC ? 0.0 : (body < SK_NB ? kc.box_invw[body] : (body == SKB_ROD ? kc.invw_rod : (body >= SKB_HAND ? kc.invw_hand : (body >= SKB_TIP ? kc.invw_tip[(body - SKB_FINGER) & 1] : kc.invw_finger[(body - SKB_FINGER) & 1])))); };
  const double R0 = fmax(1e-15, (1 - imp) / imp * (invw(a) + invw(b)));
  const double R1 = R0 / fmax(1e-15, kc.impratio);
  for (int r = 0; r < 4; r++) {
    const double v = r < R.dim ? sk_dot(sc, R, r, ST_VEL) : 0.0;
    SG(base + 16 + r) = r < R.dim ? -ps.B * v - (r == 0 ? ps.K * imp * (dist - ps.margin) : 0.0) : 0.0;
  }
  SG(base + 20) = 1 / R0; SG(base + 21) = 1 / R1; SG(base + 22) = 1 / R1;
  SG(base + 23) = 1 / (R1 * ps.fric[0] * ps.fric[0] / (ps.fric[1] * ps.fric[1]));
  SG(base + 24) = ps.fric[0] * sqrt(R1 / R0);
}
template <int OP> D3IL_HD void sk_contact_dispatch(const StackConsts& kc, const StackScratch sc, int ci, int mode) {
  const int kind = sk_kind((int)SG(ci * SREC + 13), (int)SG(ci * SREC + 14));
  if (OP == 0) {
    if (kind == 0) sk_contact_gh<0, 6>(kc, sc, ci, mode != 0); else if (kind == 1) sk_contact_gh<6, 6>(kc, sc, ci, mode != 0);
    else if (kind == 2) sk_contact_gh<6, 9>(kc, sc, ci, mode != 0); else sk_contact_gh<0, 9>(kc, sc, ci, mode != 0);
  } else {
    if (kind == 0) sk_contact_dot<0, 6>(kc, sc, ci, mode); else if (kind == 1) sk_contact_dot<6, 6>(kc, sc, ci, mode);
    else if (kind == 2) sk_contact_dot<6, 9>(kc, sc, ci, mode); else sk_contact_dot<0, 9>(kc, sc, ci, mode);
  }
}

// Cholesky of the island's part of the packed Hessian.  bm: blocks of the island; cm[i]: blocks coupled with block i (after fill
// closure).  Structurally empty blocks are skipped.
D3IL_NOINLINE inline bool sk_chol(const StackScratch sc, unsigned bm, const unsigned* cm) {
  bool ok = true;
  for (int bi = 0; bi <= SK_NB; bi++) {
    if (!((bm >> bi) & 1u)) continue;
    for (int i = sk_blk0(bi); i < sk_blk0(bi) + sk_blkn(bi); i++) {
      for (int bj = 0; bj <= bi; bj++) {
        if (!((bm >> bj) & 1u) || (bi != bj && !((cm[bi] >> bj) & 1u))) continue;
        const int jend = bi == bj ? i + 1 : sk_blk0(bj) + sk_blkn(bj);
        for (int j = sk_blk0(bj); j < jend; j++) {
          double s = SL(ST_H + tri(i, j));
          for (int bk = 0; bk <= bj; bk++) {      // columns of the blocks coupled with both rows
            if (!((bm >> bk) & 1u) || !(bk == bj || ((cm[bj] >> bk) & 1u)) || !(bk == bi || ((cm[bi] >> bk) & 1u))) continue;
            const int k0 = sk_blk0(bk), k1 = k0 + sk_blkn(bk) < j ? k0 + sk_blkn(bk) : j;
            for (int k = k0; k < k1; k++) s -= SL(ST_H + tri(i, k)) * SL(ST_H + tri(j, k));
          }
          if (i == j) {
            if (!(s > 0)) { ok = false; s = 1; }
            SL(ST_H + tri(i, i)) = sqrt(s);
          } else SL(ST_H + tri(i, j)) = s / SL(ST_H + tri(j, j));
        }
      }
    }
  }
  return ok;
}
D3IL_NOINLINE inline void sk_chol_solve(const StackScratch sc, unsigned bm, const unsigned* cm, int vec) {
  for (int bi = 0; bi <= SK_NB; bi++) {
    if (!((bm >> bi) & 1u)) continue;
    for (int i = sk_blk0(bi); i < sk_blk0(bi) + sk_blkn(bi); i++) {
      double s = SL(vec + i);
      for (int bk = 0; bk <= bi; bk++) {
        if (!((bm >> bk) & 1u) || !(bk == bi || ((cm[bi] >> bk) & 1u))) continue;
        const int k0 = sk_blk0(bk), k1 = k0 + sk_blkn(bk) < i ? k0 + sk_blkn(bk) : i;
        for (int k = k0; k < k1; k++) s -= SL(ST_H + tri(i, k)) * SL(vec + k);
      }
      SL(vec + i) = s / SL(ST_H + tri(i, i));
    }
  }
  for (int bi = SK_NB; bi >= 0; bi--) {
    if (!((bm >> bi) & 1u)) continue;
    for (int i = sk_blk0(bi) + sk_blkn(bi) - 1; i >= sk_blk0(bi); i--) {
      double s = SL(vec + i);
      for (int bk = bi; bk <= SK_NB; bk++) {
        if (!((bm >> bk) & 1u) || !(bk == bi || ((cm[bi] >> bk) & 1u))) continue;
        const int k0 = sk_blk0(bk) > i + 1 ? sk_blk0(bk) : i + 1, k1 = sk_blk0(bk) + sk_blkn(bk);
        for (int k = k0; k < k1; k++) s -= SL(ST_H + tri(k, i)) * SL(vec + k);
      }
      SL(vec + i) = s / SL(ST_H + tri(i, i));
    }
  }
}

// ------------------------------------------------------------------------------------------------ the Newton solve
// Primal Newton on ONE island (bm: its blocks) of the constraint system: x (ST_X) in: start point, out: optimum.  The contacts of the
// island are those whose (non-static) bodies lie in bm; the joint-limit rows belong to the arm block.
D3IL_NOINLINE inline bool sk_solve_island(const StackConsts& kc_, const StackScratch sc, int ncon, unsigned bm, const unsigned* cm, bool warm) {
  D3IL_STACK_CONSTS(kc_, kc);
  const bool arm = ((bm >> SK_NB) & 1u) != 0;
#define SK_FOR_DOFS(i) for (int b_ = 0; b_ <= SK_NB; b_++) if ((bm >> b_) & 1u) for (int i = sk_blk0(b_); i < sk_blk0(b_) + sk_blkn(b_); i++)
#define SK_IN_ISLAND(ci) ((bm >> sk_blk_of((int)SG((ci) * SREC + 14))) & 1u)     /* body 2 is never static */
  bool converged = false;
  SK_TIC;
  if (warm) {      // warm start: a start point that still satisfies the gradient tolerance (a resting box) is accepted after ONE gradient pass
    SK_FOR_DOFS(i) SL(ST_G + i) = sk_Mv(kc, sc, i, ST_X, ST_A0);
    if (arm)
      for (int a = 0; a < NDOF; a++) {
        const double sg = SL(ST_LIM + 3 * a), D = SL(ST_LIM + 3 * a + 1), ar = SL(ST_LIM + 3 * a + 2);
        if (sg != 0) { const double jar = sg * SL(ST_X + SK_ARM0 + a) - ar; if (jar < 0) SL(ST_G + SK_ARM0 + a) += sg * D * jar; }
      }
    for (int ci = 0; ci < ncon; ci++) if (SK_IN_ISLAND(ci)) sk_contact_dispatch<0>(kc, sc, ci, 0);
    double gm = 0;
    SK_FOR_DOFS(i) gm = fmax(gm, fabs(SL(ST_G + i)));
    SK_TOC(6);
    if (gm <= D3IL_TOL.grad_tol) return true;
  }
  for (int it = 0; it < 60 && !converged; it++) {
    // gradient and Hessian at x
    for (int bi = 0; bi <= SK_NB; bi++) if ((bm >> bi) & 1u)
      for (int i = sk_blk0(bi); i < sk_blk0(bi) + sk_blkn(bi); i++)
        for (int bj = 0; bj <= bi; bj++) if ((bm >> bj) & 1u) {
          const int jend = bi == bj ? i + 1 : sk_blk0(bj) + sk_blkn(bj);
          for (int j = sk_blk0(bj); j < jend; j++) SL(ST_H + tri(i, j)) = 0;
        }
    SK_FOR_DOFS(i) SL(ST_G + i) = sk_Mv(kc, sc, i, ST_X, ST_A0);
    for (int b = 0; b < SK_NB; b++) if ((bm >> b) & 1u) for (int k = 0; k < 6; k++) SL(ST_H + tri(6 * b + k, 6 * b + k)) = k < 3 ? kc.box_mass[b] : kc.box_inertia[b][k - 3];
    if (arm) {
      for (int a = 0; a < NDOF; a++) for (int k = 0; k <= a; k++) SL(ST_H + tri(SK_ARM0 + a, SK_ARM0 + k)) = SL(ST_M + tri(a, k));
      for (int a = 0; a < NDOF; a++) {     // joint-limit rows
        const double sg = SL(ST_LIM + 3 * a), D = SL(ST_LIM + 3 * a + 1), ar = SL(ST_LIM + 3 * a + 2);
        if (sg != 0) {
          const double jar = sg * SL(ST_X + SK_ARM0 + a) - ar;
          if (jar < 0) { SL(ST_G + SK_ARM0 + a) += sg * D * jar; SL(ST_H + tri(SK_ARM0 + a, SK_ARM0 + a)) += D; }
        }
      }
    }
    SK_TOC(7);
    for (int ci = 0; ci < ncon; ci++) if (SK_IN_ISLAND(ci)) sk_contact_dispatch<0>(kc, sc, ci, 1);
    SK_TOC(8);
    double gm = 0;
    SK_FOR_DOFS(i) gm = fmax(gm, fabs(SL(ST_G + i)));
    SG(SG_DIAG) += 1; SG(SG_DIAG + 1) = fmax(SG(SG_DIAG + 1), gm);
    if (gm <= D3IL_TOL.grad_tol) { converged = true; break; }
    if (!sk_chol(sc, bm, cm)) return false;
    SK_TOC(9);
    SK_FOR_DOFS(i) SL(ST_P + i) = -SL(ST_G + i);
    sk_chol_solve(sc, bm, cm, ST_P);
    SK_TOC(10);
    // line search: phi'(alpha) = p' M (x - a0) + alpha p' M p - sum f(jar + alpha Jp) . Jp, safeguarded Newton on alpha
    double pMp = 0, pMa = 0, gTp = 0;
    SK_FOR_DOFS(i) gTp += SL(ST_G + i) * SL(ST_P + i);
    for (int b = 0; b < SK_NB; b++) if ((bm >> b) & 1u)
      for (int k = 0; k < 6; k++) { const int i = 6 * b + k; const double m = k < 3 ? kc.box_mass[b] : kc.box_inertia[b][k - 3], p = SL(ST_P + i); pMp += m * p * p; pMa += m * p * (SL(ST_X + i) - SL(ST_A0 + i)); }
    if (arm)
      for (int a = 0; a < NDOF; a++) {
        double mp = 0, ma = 0;
        for (int k = 0; k < NDOF; k++) { const double m = SL(ST_M + (a >= k ? tri(a, k) : tri(k, a))); mp += m * SL(ST_P + SK_ARM0 + k); ma += m * (SL(ST_X + SK_ARM0 + k) - SL(ST_A0 + SK_ARM0 + k)); }
        pMp += SL(ST_P + SK_ARM0 + a) * mp; pMa += SL(ST_P + SK_ARM0 + a) * ma;
      }
    for (int ci = 0; ci < ncon; ci++) if (SK_IN_ISLAND(ci)) sk_contact_dispatch<1>(kc, sc, ci, 0);
    SK_TOC(11);
    double alpha = 1, lo = 0, hi = -1, best = 1, wprev = 1e300;
    for (int ls = 0; ls < 50; ls++) {
      double d1 = pMa + alpha * pMp, d2 = pMp;
      if (arm)
        for (int a = 0; a < NDOF; a++) {
          const double sg = SL(ST_LIM + 3 * a), D = SL(ST_LIM + 3 * a + 1), ar = SL(ST_LIM + 3 * a + 2);
          if (sg != 0) {
            const double jp = sg * SL(ST_P + SK_ARM0 + a), jar = sg * SL(ST_X + SK_ARM0 + a) - ar + alpha * jp;
            if (jar < 0) { d1 += D * jar * jp; d2 += D * jp * jp; }
          }
        }
      for (int ci = 0; ci < ncon; ci++) if (SK_IN_ISLAND(ci)) {
        const int base = ci * SREC, set = (int)SG(base + 15), dim = kc.set[set].dim;
        double jt[4], jp[4], D[4], f[4], Hc[16], fr[3];
#pragma unroll
        for (int r = 0; r < 4; r++) { jp[r] = SG(base + 29 + r); jt[r] = SG(base + 25 + r) + alpha * jp[r]; D[r] = SG(base + 20 + r); }
        sk_row_fric(kc.set[set], fr);
        sk_cone(dim, jt, D, SG(base + 24), fr, f, Hc);
#pragma unroll
        for (int r = 0; r < 4; r++) { d1 -= f[r] * jp[r];
#pragma unroll
          for (int q = 0; q < 4; q++) d2 += jp[r] * Hc[4 * r + q] * jp[q]; }
      }
      best = alpha;
      if (ls == 0 && d1 <= D3IL_TOL.ls_full * fabs(gTp)) break;
      if (fabs(d1) <= D3IL_TOL.ls_c2 * fabs(gTp) || fabs(d1) <= D3IL_TOL.ls_rel * d2 * alpha || fabs(d1) < 1e-14 * fmax(1.0, fabs(pMa))) break;
      if (d1 < 0) lo = alpha; else hi = alpha;
      double na = alpha - d1 / d2;
      if (hi >= 0) {
        const double wbr = hi - lo;
        const bool slow = wbr > 0.5 * wprev;
        wprev = wbr;
        if (slow || !(na > lo && na < hi)) na = 0.5 * (lo + hi);
      } else if (na <= lo) na = 2 * lo + 1;
      if (na == alpha) break;
      alpha = na;
    }
    SK_TOC(12);
    double smax = 0, xmax = 0;
    SK_FOR_DOFS(i) { const double dx = best * SL(ST_P + i); SL(ST_X + i) += dx; smax = fmax(smax, fabs(dx)); xmax = fmax(xmax, fabs(SL(ST_X + i))); }
    if (smax <= 1e-12 * (1 + xmax) || (best == 1.0 && smax <= D3IL_TOL.step_rel * (1 + xmax))) converged = true;
  }
#undef SK_FOR_DOFS
#undef SK_IN_ISLAND
  return converged;
}
// All islands of the sub-step: connected components of {box 0, box 1, box 2, arm} under the contacts.  Blocks without any
// constraint keep x = a0.
D3IL_HD bool sk_solve(const StackConsts& kc, const StackScratch sc, int ncon, bool any_lim, bool warm) {
  unsigned adj[SK_NB + 1] = {0, 0, 0, 0}, has = 0;
  for (int ci = 0; ci < ncon; ci++) {
    const int ba = sk_blk_of((int)SG(ci * SREC + 13)), bb = sk_blk_of((int)SG(ci * SREC + 14));
    has |= 1u << bb;
    if (ba >= 0) { has |= 1u << ba; if (ba != bb) { adj[ba] |= 1u << bb; adj[bb] |= 1u << ba; } }
  }
  if (any_lim) has |= 1u << SK_NB;
  bool ok = true;
  unsigned done = 0;
  for (int b0 = 0; b0 <= SK_NB; b0++) {
    if (!((has >> b0) & 1u) || ((done >> b0) & 1u)) continue;
    unsigned bm = 1u << b0;
    for (int rep = 0; rep <= SK_NB; rep++) for (int b = 0; b <= SK_NB; b++) if ((bm >> b) & 1u) bm |= adj[b];
    done |= bm;
    unsigned cm[SK_NB + 1];
    for (int b = 0; b <= SK_NB; b++) cm[b] = adj[b] & bm;
    for (int k = 0; k <= SK_NB; k++)       // fill closure: eliminating block k couples every pair of later blocks it touches
      for (int i = k + 1; i <= SK_NB; i++) if ((cm[k] >> i) & 1u)
        for (int j = k + 1; j <= SK_NB; j++) if (j != i && ((cm[k] >> j) & 1u)) cm[i] |= 1u << j;
    ok = sk_solve_island(kc, sc, ncon, bm, cm, warm) && ok;
  }
  return ok;
}

#if defined(__HIPCC__)
// ------------------------------------------------------------------------------------------------ wave-cooperative Newton solve
// The same problem as sk_solve, solved for TWO environments at a time by the 64 lanes of their wave, one per half wave (sk_solve_dual below):
//   * lane c < ncon owns contact c: frame, wrench arms, reference accelerations / regularisation / residuals in the lane's registers (contacts in
//     wrench form since round 6: no constraint rows); the passes over the contacts (pair matrices and wrenches for gradient + Hessian, J p, every
//     trial of the line search) run on all contacts at once, sums through LDS additions into body-pair slots or half-wave reductions;
//   * lane i < 27 owns dof i: its motion-subspace column, its gradient entry, its row of the Hessian and of the Cholesky factor (registers, columns
//     broadcast inside the half, all loops unrolled over the 27 x 27 lower triangle), its entry of the search direction;
//   * no islands: the block-diagonal system is factorised as a whole (MuJoCo 2.3.2 solves it as a whole too); blocks without
//     constraints start at their smooth acceleration with zero gradient and do not move.
// Same stopping rule, line search and tolerances as sk_solve_island.
__device__ __forceinline__ double sk_bcast(double v, int src /* wave-uniform */) {
  unsigned long long u; __builtin_memcpy(&u, &v, 8);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, src), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), src);
  const unsigned long long r = ((unsigned long long)hi << 32) | lo; double out; __builtin_memcpy(&out, &r, 8); return out;
}
// Convergence fence.  Every cross-lane sequence of this engine - a DPP butterfly, a ds_bpermute shuffle - is bracketed by
// a side-effecting convergent no-op (llvm.amdgcn.wave.barrier: no instruction is emitted).  Without one between the group reduction and the
// shuffle of sk_support1_group_pre, hipcc (ROCm 7.2) builds a kernel whose MPR results depend on the lane group a job runs on (the
// -DD3IL_SK_PRELOAD_RAW build).  What the round-4 experiments established: the defect is deterministic; it survives every register
// allocator (greedy / basic for VGPRs, SGPRs, WWM), the SLP vectoriser on or off, and waits / nops inserted into its ISA after every LDS,
// DPP, lane, EXEC-writing and transcendental instruction (so: neither allocation nor a hardware hazard); making the shuffled values opaque
// WITHOUT a side effect (a non-volatile asm) does not remove it, a fence that leaves them transparent does; with SimplifyCFG's
// common-code sinking off (-mllvm -simplifycfg-sink-common=false) 1913 of the 1917 deviating environments of the probe disappear.  I.e. the
// optimiser restructures the divergent if-chains around the group-wide operations unless a side effect pins them.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(D3IL_SK_NO_FENCE) && !defined(D3IL_SK_PRELOAD_RAW)
#define SK_CONVERGE() __builtin_amdgcn_wave_barrier()
#else
#define SK_CONVERGE() ((void)0)
#endif
// Reductions over the 64 lanes (all active): butterfly inside the rows of 16 lanes with DPP moves (quad permutes, half-row and row
// mirrors: no LDS crossbar round trips), then the four row results through v_readlane.  Every lane gets the result.
__device__ __forceinline__ double sk_dpp_mov(double v, const int ctrl_sel) {
  unsigned long long u; __builtin_memcpy(&u, &v, 8);
  int lo = (int)(unsigned)u, hi = (int)(unsigned)(u >> 32);
  switch (ctrl_sel) {      // compile-time selector -> immediate DPP control
    case 0: lo = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xF, 0xF, false); break;    // quad_perm [1,0,3,2]
    case 1: lo = __builtin_amdgcn_update_dpp(lo, lo, 0x4E, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x4E, 0xF, 0xF, false); break;    // quad_perm [2,3,0,1]
    case 2: lo = __builtin_amdgcn_update_dpp(lo, lo, 0x141, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x141, 0xF, 0xF, false); break;  // row_half_mirror
    default: lo = __builtin_amdgcn_update_dpp(lo, lo, 0x140, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x140, 0xF, 0xF, false); break; // row_mirror
  }
  const unsigned long long r = ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo; double out; __builtin_memcpy(&out, &r, 8); return out;
}
__device__ __forceinline__ double sk_wave_sum(double v) {
  SK_CONVERGE();
  v += sk_dpp_mov(v, 0); v += sk_dpp_mov(v, 1); v += sk_dpp_mov(v, 2); v += sk_dpp_mov(v, 3);      // every lane: the sum of its row of 16
  const double r = (sk_bcast(v, 0) + sk_bcast(v, 16)) + (sk_bcast(v, 32) + sk_bcast(v, 48));
  SK_CONVERGE();
  return r;
}
__device__ __forceinline__ double sk_wave_max(double v) {
  SK_CONVERGE();
  v = fmax(v, sk_dpp_mov(v, 0)); v = fmax(v, sk_dpp_mov(v, 1)); v = fmax(v, sk_dpp_mov(v, 2)); v = fmax(v, sk_dpp_mov(v, 3));
  const double r = fmax(fmax(sk_bcast(v, 0), sk_bcast(v, 16)), fmax(sk_bcast(v, 32), sk_bcast(v, 48)));
  SK_CONVERGE();
  return r;
}
constexpr int SKC_JSIZE = 4 * 15 * SK_MAXCON;   // doubles per wave of the shared W area (rounds 3 - 5: the contact rows, 4 x 15 per contact; now the wrench tables, SKW_*)
// LDS layout of the step kernel (one wave = one workgroup = SK_LANES environments):
//   shared by the workgroup : two solver heads (H [SK_NH] packed Hessian / Cholesky factor | G gradient | P direction) - the constraint
//                             solver works on TWO environments at a time, one per half wave - | W [SKC_JSIZE] their wrench tables (SKW_*: motion-
//                             subspace columns, twists, body-pair slots, aggregates) and the workgroup's carried sin / cos table.  During the
//                             collision phase: per-lane staging of the contacts a pair test emits.
//   per environment         : the t area of the one-lane code WITHOUT its head (vectors, mass matrix, kinematic tables, limit rows; the
//                             view pointer is shifted by ST_HEAD so that the ST_* offsets stay valid), then the part of the state that is
//                             not in a table already (q, qfrc_bias, TCP, box quaternions - velocities and box positions live in ST_VEL /
//                             ST_BP), the action, the contact count / flags words and the COMPACT contact records (8 doubles: position,
//                             normal, distance, bodies | parameter set - the tangents are re-derived with make_frame).
// Nothing of an environment lives in registers between the phases of a sub-step and nothing goes through HBM inside the step.
constexpr int SREC2 = 8;
constexpr int SE_Q = ST_SIZE, SE_BIAS = SE_Q + NDOF, SE_TCP = SE_BIAS + NARM, SE_BQ = SE_TCP + 3, SE_ACT = SE_BQ + 4 * SK_NB;
constexpr int SE_NCON = SE_ACT + NARM;          // contact count of this sub-step
constexpr int SE_NEED = SE_NCON + 1;            // 1: constraints present (contacts or joint limits) -> the solver runs; + 256: contacts were dropped
constexpr int SE_REC = SE_NEED + 1;
constexpr int SE_END = SE_REC + SK_MAXCON * SREC2;
constexpr int SE_SIZE = ((SE_END - ST_HEAD) | 1);  // doubles per environment (odd: the environments start on different banks)
constexpr int SKC_SHARED = 2 * ST_HEAD + SKC_JSIZE;
constexpr int SKC_STAGE = 36;                   // staging doubles per lane in the shared area: normal[3] + 8 x (dist, pos[3]) + count
static_assert(SKC_STAGE * 16 * SK_LANES <= SKC_SHARED, "contact staging must fit the shared area");
static_assert(SK_LANES % 2 == 0, "the solver takes the environments of a workgroup in pairs");
__device__ __forceinline__ sk_lds_double* sk_env_view(sk_lds_double* smem, int e) { return smem + SKC_SHARED + e * SE_SIZE - ST_HEAD; }
// Contacts in wrench form (round 6).  The constraint rows of a contact are never materialised.  With spatial quantities about the fixed
// reference point SKW_REF (angular part first), the row r of a contact at p with direction d_r is the wrench W_r = [(p - ref) x d_r ; d_r]
// (the torsional row: [n ; 0]), every dof i has the motion-subspace column s_i (box: linear [0 ; e_k], angular [R e_k ; (c - ref) x R e_k];
// arm joint k: [z_k ; (o_k - ref) x z_k]; finger slide: [0 ; axis]), and J_r[i] = +/- W_r . s_i for the dofs that move body 2 / body 1
// (sk_box_rows / sk_arm_rows written out).  So
//   J x      = W . (T_B - T_A),  T_body = sum of s_i x_i over the dofs that move the body (six twists per environment and vector),
//   J' f     = +/- s_i . F,      F = sum_r f_r W_r (one 6-vector per contact),
//   J' Hc J  = s_i' K s_j,       K = W' Hc W (one symmetric 6 x 6 per contact),
// and contacts of the same body pair ADD in F and K before anything is projected onto the dofs: a contact lane issues 27 LDS additions
// per pass (21 + 6 into the slot of its body pair) instead of the 15 + 120 of a box <-> finger row block, the dof lanes build their Hessian
// row in registers from the (aggregated) pair matrices - H[i][j] = s_i' Keff(class i, class j) s_j -, and the J area of rounds 3 - 5 is gone.
// LDS of one half wave (its environment) inside the workgroup's shared W area:
constexpr int SKW_HALF = SKC_JSIZE / 2;
constexpr int SKW_KST = 22;                           // a packed symmetric 6 x 6 (21 entries) + pad: tables start on 16-byte boundaries
constexpr int SKW_SCOL = 0;                           // s_i, 27 x 6
constexpr int SKW_TW = SKW_SCOL + 6 * SK_NV;          // twists of the generalised bodies box 0 | 1 | 2 | hand (joints 0 .. 6) | finger 0 | finger 1
constexpr int SKW_SLOTK = SKW_TW + 36;                // K of the body pairs: static-box b (b) | box-box 01 02 12 (3 ..) | box-hand (6 + b) | box-finger (9 + 2 b + f) | finger-finger (15)
constexpr int SKW_SLOTF = SKW_SLOTK + 16 * SKW_KST;   // F of the body pairs, oriented body 1 -> body 2 of the slot
constexpr int SKW_AGGK = SKW_SLOTF + 16 * 6;          // Kd_b (all pairs of box b) [3] | K_b,arm [3] | Kd'_f (box pairs of finger f) [2] | Kd_f = Kd'_f + FF [2] | K_rev
constexpr int SKW_AGGN = SKW_AGGK + 11 * SKW_KST;     // net wrench per dof class: box 0 | 1 | 2 | joints 0 .. 6 | slide 0 | slide 1
constexpr int SKW_END = SKW_AGGN + 36;
// the upper half's tables follow the lower half's directly; behind them (and behind everything the collision phase stages) the sin / cos of the arm joints of
// the workgroup's environments, carried from sub-step to sub-step (trig_advance) instead of seven sincos evaluations per environment and sub-step
constexpr int SKW_TRIG = 2 * SKW_END;                  // offset in the W area: SK_LANES x 14 doubles
static_assert(SKW_TRIG + 2 * NARM * SK_LANES <= SKC_JSIZE, "the wrench tables of two environments and the trig table must fit the shared area");
static_assert(2 * ST_HEAD + SKW_TRIG >= SKC_STAGE * 16 * SK_LANES, "the trig table must lie behind the collision phase's staging area");
__device__ constexpr double SKW_REF[3] = {0.5, 0.0, 0.0};
__device__ __forceinline__ int sk_tri6(int a, int b) { return a >= b ? a * (a + 1) / 2 + b : b * (b + 1) / 2 + a; }
__device__ __forceinline__ int sk_genbody(int body) { return body < SK_NB ? body : (sk_finger_of(body) < 0 ? 3 : 4 + sk_finger_of(body)); }

#if defined(D3IL_DEVICE_STATS)
#define D3IL_SD_COUNT(slot) atomicAdd(&g_dev_stats[16 + (slot)], 1ull)
#else
#define D3IL_SD_COUNT(slot) ((void)0)
#endif
// Half-wave helpers: lanes 0 .. 31 work on one environment, lanes 32 .. 63 on another.
__device__ __forceinline__ double sk_hbcast(double v, int j /* wave-uniform, 0 .. 31 */, bool upper) {
#if defined(D3IL_SK_HBCAST_READLANE)
  const double a = sk_bcast(v, j), b = sk_bcast(v, 32 + j);
  return upper ? b : a;
#else
  // lane j of the caller's own half through the LDS crossbar (ds_bpermute_b32 x 2): one instruction per word for BOTH halves, and the
  // independent broadcasts of a factorisation step pipeline - v_readlane would need two reads + a select per word and half
  return __shfl(v, (upper ? 32 : 0) | j);
#endif
}
// the same value through v_readlane (two reads per word + a select): few cycles of latency instead of an LDS round trip - for the broadcasts the
// next instruction of a dependent chain waits for (pivot of a factorisation column, the entry a substitution step hands on)
__device__ __forceinline__ double sk_hbcast_chain(double v, int j /* compile-time, 0 .. 31 */, bool upper) {
  const double a = sk_bcast(v, j), b = sk_bcast(v, 32 + j);
  return upper ? b : a;
}
__device__ __forceinline__ double sk_rsqrt1(double x) {      // 1 / sqrt(x) with ONE Newton step: for the factor's pivots, whose rounding only shapes the search direction
  double y = __builtin_amdgcn_rsq(x);
  return y * (1.5 - 0.5 * x * y * y);
}
__device__ __forceinline__ double sk_half_sum(double v, bool upper) {
  SK_CONVERGE();
  v += sk_dpp_mov(v, 0); v += sk_dpp_mov(v, 1); v += sk_dpp_mov(v, 2); v += sk_dpp_mov(v, 3);      // every lane: the sum of its row of 16
  const double a = sk_bcast(v, 0) + sk_bcast(v, 16), b = sk_bcast(v, 32) + sk_bcast(v, 48);
  SK_CONVERGE();
  return upper ? b : a;
}
__device__ __forceinline__ double sk_half_max(double v, bool upper) {
  SK_CONVERGE();
  v = fmax(v, sk_dpp_mov(v, 0)); v = fmax(v, sk_dpp_mov(v, 1)); v = fmax(v, sk_dpp_mov(v, 2)); v = fmax(v, sk_dpp_mov(v, 3));
  const double a = fmax(sk_bcast(v, 0), sk_bcast(v, 16)), b = fmax(sk_bcast(v, 32), sk_bcast(v, 48));
  SK_CONVERGE();
  return upper ? b : a;
}
// The constraint problems of TWO environments (e0 on the lower half wave, e0 + 1 on the upper one; act0 / act1: which of them is solved),
// same algorithm, stopping rule, line search and tolerances as sk_solve_island.  Per half: lane hl < ncon owns contact hl (frame, wrench
// arms, reference accelerations / regularisation / residuals in its registers), lane hl < 27 owns dof hl (its column s_i, gradient entry,
// row of the Hessian and of the Cholesky factor in registers, columns broadcast inside the half).  The two halves iterate in lock step
// until both have finished; a finished half idles.
// Returns bit 0 / bit 1: the solve of the lower / upper half FAILED (non-positive pivot or iteration cap).
__device__ __forceinline__ unsigned sk_solve_dual(const StackConsts& kc_, sk_lds_double* smem, const int e0, const int lane, const bool act0, const bool act1) {
  D3IL_STACK_CONSTS(kc_, kc);
  const bool upper = lane >= 32;
  const int hl = lane & 31;
  sk_lds_double* const t = sk_env_view(smem, e0 + (upper ? 1 : 0));
  sk_lds_double* const Hs = smem + (upper ? ST_HEAD : 0);
  sk_lds_double* const Ps = Hs + ST_P;
  sk_lds_double* const Wb = smem + 2 * ST_HEAD + (upper ? SKW_END : 0);
  const bool active = upper ? act1 : act0;
  const int ncon = active ? (int)t[SE_NCON] : 0;
  const bool con = hl < ncon;
  const int i = hl;                         // dof owned by this lane (within its half)
  const bool row = active && hl < SK_NV, armrow = row && hl >= SK_ARM0;
  const int ia = armrow ? hl - SK_ARM0 : 0;
  const int bi = hl >= SK_ARM0 ? SK_NB : hl / 6;      // block of the dof: box 0 | 1 | 2 | arm
  const bool slide = armrow && ia >= NARM;
  const int fs = slide ? ia - NARM : 0;
#if defined(D3IL_DEVICE_STATS)
  unsigned long long sd_t0 = wall_clock64();
#define SD_TOC(slot) do { unsigned long long t_ = wall_clock64(); if (lane == 0 && blockIdx.x == 0) atomicAdd(&g_dev_stats[16 + (slot)], t_ - sd_t0); sd_t0 = t_; } while (0)
  if (lane == 0 && blockIdx.x == 0) atomicAdd(&g_dev_stats[16 + 6], 1ull);
#else
#define SD_TOC(slot) ((void)0)
#endif
  // ---- dof lanes: motion-subspace column s_i (registers + LDS)
  double si[6] = {0, 0, 0, 0, 0, 0};
  if (row) {
    if (!armrow) {
      const int b = hl / 6, k = hl - 6 * b;
      if (k < 3) { si[3] = k == 0 ? 1.0 : 0.0; si[4] = k == 1 ? 1.0 : 0.0; si[5] = k == 2 ? 1.0 : 0.0; }
      else {
        const double w[3] = {t[ST_BR + 9 * b + (k - 3)], t[ST_BR + 9 * b + 3 + (k - 3)], t[ST_BR + 9 * b + 6 + (k - 3)]};      // body axis k - 3 in the world frame
        const double c[3] = {t[ST_BP + 3 * b] - SKW_REF[0], t[ST_BP + 3 * b + 1] - SKW_REF[1], t[ST_BP + 3 * b + 2] - SKW_REF[2]};
        si[0] = w[0]; si[1] = w[1]; si[2] = w[2];
        si[3] = c[1] * w[2] - c[2] * w[1]; si[4] = c[2] * w[0] - c[0] * w[2]; si[5] = c[0] * w[1] - c[1] * w[0];
      }
    } else if (!slide) {
      const double z[3] = {t[ST_Z + 3 * ia], t[ST_Z + 3 * ia + 1], t[ST_Z + 3 * ia + 2]};
      const double o[3] = {t[ST_O + 3 * ia] - SKW_REF[0], t[ST_O + 3 * ia + 1] - SKW_REF[1], t[ST_O + 3 * ia + 2] - SKW_REF[2]};
      si[0] = z[0]; si[1] = z[1]; si[2] = z[2];
      si[3] = o[1] * z[2] - o[2] * z[1]; si[4] = o[2] * z[0] - o[0] * z[2]; si[5] = o[0] * z[1] - o[1] * z[0];
    } else if (kc.variant == SKV_STACKING) { si[3] = t[ST_FAX + 3 * fs]; si[4] = t[ST_FAX + 3 * fs + 1]; si[5] = t[ST_FAX + 3 * fs + 2]; }
    // (the rod-robot variants have no finger geoms: stack_pre_kin writes no slide axes there, no contact touches a finger body, the column stays zero -
    // reading the unwritten table words made the slide rows of the Hessian depend on what the LDS held before: NaN on some boxes of the pool, found with the
    // poison build, profiles/r06/aligning_uninitialised_lds/)
#pragma unroll
    for (int a = 0; a < 6; a++) Wb[SKW_SCOL + 6 * i + a] = si[a];
  }
  // ---- contact lanes: frame, wrench arms, slot of the body pair, twist addresses
  double cd[3][3] = {{0, 0, 1}, {1, 0, 0}, {0, 1, 0}}, cm[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};      // d_r (normal, tangent 1, tangent 2), m_r = (p - ref) x d_r
  int tA = -1, tB = SKW_TW, slot = 0, cdim = 3, cset = 0, cba = -1, cbb = 0, cbodya = SKB_STATIC, cbodyb = 0;
  double sgF = 1, cdist = 0;
  if (con) {
    const sk_lds_double* r = t + SE_REC + hl * SREC2;
    const double p[3] = {r[0] - SKW_REF[0], r[1] - SKW_REF[1], r[2] - SKW_REF[2]};
    cd[0][0] = r[3]; cd[0][1] = r[4]; cd[0][2] = r[5];
    make_frame(cd[0], cd[1], cd[2]);
#pragma unroll
    for (int q = 0; q < 3; q++) { cm[q][0] = p[1] * cd[q][2] - p[2] * cd[q][1]; cm[q][1] = p[2] * cd[q][0] - p[0] * cd[q][2]; cm[q][2] = p[0] * cd[q][1] - p[1] * cd[q][0]; }
    cdist = r[6];
    const int meta = (int)r[7];
    cbodya = meta & 15; cbodyb = (meta >> 4) & 15; cset = meta >> 8;
    cdim = kc.set[cset].dim;
    cba = sk_blk_of(cbodya); cbb = sk_blk_of(cbodyb);
    const int ga = cbodya == SKB_STATIC ? -1 : sk_genbody(cbodya), gb = sk_genbody(cbodyb);
    tB = SKW_TW + 6 * gb; tA = ga < 0 ? -1 : SKW_TW + 6 * ga;
    if (ga < 0) slot = gb;                                                        // static <-> box
    else if (ga < 3 && gb < 3) { slot = 3 + ga + gb - 1; sgF = ga < gb ? 1.0 : -1.0; }      // box <-> box, stored with the lower box as body 1
    else if (ga < 3) slot = gb == 3 ? 6 + ga : 9 + 2 * ga + (gb - 4);            // box <-> hand / rod, box <-> finger
    else if (gb < 3) { slot = ga == 3 ? 6 + gb : 9 + 2 * gb + (ga - 4); sgF = -1.0; }
    else { slot = 15; sgF = (ga == 5 && gb == 4) ? -1.0 : 1.0; }                  // finger <-> finger, stored finger 0 -> finger 1
  }
  // ---- rows of the block-diagonal mass matrix and the joint-limit row of this lane's dof
  double mdiag = 0, Ma[NDOF];
#pragma unroll
  for (int k = 0; k < NDOF; k++) Ma[k] = 0;
  if (row && !armrow) { const int b = i / 6, k = i - 6 * b; mdiag = k < 3 ? kc.box_mass[b] : kc.box_inertia[b][k - 3]; }
  if (armrow) {
#pragma unroll
    for (int k = 0; k < NDOF; k++) Ma[k] = t[ST_M + (ia >= k ? tri(ia, k) : tri(k, ia))];
  }
  double lsg = 0, lD = 0, lar = 0;
  if (armrow) { lsg = t[ST_LIM + 3 * ia]; lD = t[ST_LIM + 3 * ia + 1]; lar = t[ST_LIM + 3 * ia + 2]; }
  bool fin = !active;      // this half has finished (converged, failed, or nothing to do); half-uniform
  bool okh = true;
  // twists of the six generalised bodies for a dof vector v (LDS): lane (body, component) of 24 sums s_i[component] v_i over the body's dofs
  auto twists = [&](const sk_lds_double* v) {
    if (hl < 24 && !fin) {
      const int body = hl / 6, comp = hl - 6 * body, i0 = body < SK_NB ? 6 * body : SK_ARM0;
      double s = 0;
#pragma unroll
      for (int k = 0; k < NARM; k++) if (k < 6 || body == SK_NB) s += Wb[SKW_SCOL + 6 * (i0 + k) + comp] * v[i0 + k];
      Wb[SKW_TW + 6 * body + comp] = s;
      if (body == SK_NB) {
        Wb[SKW_TW + 24 + comp] = s + Wb[SKW_SCOL + 6 * (SK_ARM0 + NARM) + comp] * v[SK_ARM0 + NARM];
        Wb[SKW_TW + 30 + comp] = s + Wb[SKW_SCOL + 6 * (SK_ARM0 + NARM + 1) + comp] * v[SK_ARM0 + NARM + 1];
      }
    }
  };
  // W . (T_B - T_A) of this lane's contact with the twists in the table
  auto wdot = [&](double* out) {
    double dw[6];
#pragma unroll
    for (int a = 0; a < 6; a++) dw[a] = Wb[tB + a] - (tA >= 0 ? Wb[tA + a] : 0.0);
#pragma unroll
    for (int q = 0; q < 3; q++) out[q] = cm[q][0] * dw[0] + cm[q][1] * dw[1] + cm[q][2] * dw[2] + cd[q][0] * dw[3] + cd[q][1] * dw[4] + cd[q][2] * dw[5];
    out[3] = cdim > 3 ? cd[0][0] * dw[0] + cd[0][1] * dw[1] + cd[0][2] * dw[2] : 0.0;
  };
  __syncthreads();
  twists(t + ST_VEL);
  __syncthreads();
  // ---- contacts: reference accelerations, regularisation (mj_makeImpedance for an elliptic contact [ext]; sk_contact_dot, mode 1)
  double aref[4] = {0, 0, 0, 0}, cD[4] = {1, 1, 1, 1}, cmu = 1, cimu = 0.5, cfr[3] = {1, 1, 1};
  if (con) {
    const StackSet& ps = kc.set[cset];
    const double imp = impedance(ps.solimp, cdist - ps.margin);
    auto invw = [&](int body) { return body == SKB_STATIC ? 0.0 : (body < SK_NB ? kc.box_invw[body] : (body == SKB_ROD ? kc.invw_rod : (body >= SKB_HAND ? kc.invw_hand : (body >= SKB_TIP ? kc.invw_tip[(body - SKB_FINGER) & 1] : kc.invw_finger[(body - SKB_FINGER) & 1])))); };
    const double R0 = fmax(1e-15, (1 - imp) / imp * (invw(cbodya) + invw(cbodyb)));
    const double R1 = R0 / fmax(1e-15, kc.impratio);
    double jv[4];
    wdot(jv);
#pragma unroll
    for (int r = 0; r < 4; r++) aref[r] = r < cdim ? -ps.B * jv[r] - (r == 0 ? ps.K * imp * (cdist - ps.margin) : 0.0) : 0.0;
    if (kc.variant == SKV_ALIGNING && (cbodya == 0 || cbodyb == 0)) {
      // MuJoCo's row residual is J_o qacc_o - aref with the free body's ORIGIN acceleration; this engine solves for the centre-of-mass acceleration
      // a_c = a_o + alpha x R c + w x (w x R c), so J_o qacc_o = J_c qacc_c -/+ d . (w x (w x R c)) for the body as geom 2 / geom 1 of the pair:
      // the velocity-dependent term moves into the reference acceleration
      const double sgn = cbodyb == 0 ? 1.0 : -1.0;
#pragma unroll
      for (int r = 0; r < 3; r++)
        aref[r] += sgn * (cd[r][0] * t[ST_TIPR + SV_CEN] + cd[r][1] * t[ST_TIPR + SV_CEN + 1] + cd[r][2] * t[ST_TIPR + SV_CEN + 2]);
    }
    cD[0] = 1 / R0; cD[1] = 1 / R1; cD[2] = 1 / R1; cD[3] = 1 / (R1 * ps.fric[0] * ps.fric[0] / (ps.fric[1] * ps.fric[1]));
    cmu = ps.fric[0] * sqrt(R1 / R0);
    cimu = 1.0 / fmax(1e-15, cmu * cmu * (1 + cmu * cmu));
    sk_row_fric(ps, cfr);
  }
  SD_TOC(3);
  auto m_times = [&](const sk_lds_double* va, const sk_lds_double* vb) -> double {      // (M (v_a - v_b))_i
    if (!row) return 0.0;
    if (!armrow) return mdiag * (va[i] - (vb ? vb[i] : 0.0));
    double sum = 0;
#pragma unroll
    for (int k = 0; k < NDOF; k++) sum += Ma[k] * (va[SK_ARM0 + k] - (vb ? vb[SK_ARM0 + k] : 0.0));
    return sum;
  };
  // Block structure, union over the two halves: blocks box 0 | 1 | 2 | arm.  cpl0 bit (4 bi + bk), bk <= bi: a contact touches both blocks
  // (bk == bi: the block has a contact at all); cpl: the same below the diagonal after fill closure = the structure of the Cholesky factor
  unsigned cpl0 = 0, cpl = 0;
  {
#pragma unroll
    for (int bq = 0; bq <= SK_NB; bq++) {
      if (__any(con && (cba == bq || cbb == bq))) cpl0 |= 1u << (5 * bq);
#pragma unroll
      for (int bk = 0; bk < bq; bk++) if (__any(con && ((cba == bk && cbb == bq) || (cba == bq && cbb == bk)))) cpl0 |= 1u << (4 * bq + bk);
    }
    cpl = cpl0;
#pragma unroll
    for (int k = 0; k < SK_NB; k++) {
#pragma unroll
      for (int bq = k + 1; bq <= SK_NB; bq++) {
#pragma unroll
        for (int bj = k + 1; bj < bq; bj++) if (((cpl >> (4 * bq + k)) & 1u) && ((cpl >> (4 * bj + k)) & 1u)) cpl |= 1u << (4 * bq + bj);
      }
    }
  }
  const bool has_ff = __any(con && slot == 15);
  double jar[4] = {0, 0, 0, 0}, gi = 0, xi = row ? t[ST_X + i] : 0.0;
  double Hr[SK_NV];
  bool limact = false;      // this lane's joint-limit row is active at x
  // row residuals J x - aref and M (x - a0) at the start point; every Newton step then moves them along the step (J p and M p are at hand from the
  // line search), so a pass neither forms the twists of x nor reads x again.  The pair slots start at zero; a pass clears them behind its last reader.
  __syncthreads();
  if (!fin) {
#pragma unroll
    for (int q = 0; q < (16 * SKW_KST + 16 * 6 + 31) / 32; q++) { const int w = hl + 32 * q; if (w < 16 * SKW_KST + 16 * 6) Wb[SKW_SLOTK + w] = 0; }
  }
  twists(t + ST_X);
  double mxa = m_times(t + ST_X, t + ST_A0);
  __syncthreads();
  if (con) {
    wdot(jar);
#pragma unroll
    for (int r = 0; r < 4; r++) jar[r] -= aref[r];
  }
  SD_TOC(0);
  // gradient and Hessian at x: g -> gi, H row -> Hr (registers of the dof lane); returns max |g| of the half
  auto grad_pass = [&]() -> double {
    if (con && !fin) {
      double f[4], Hc[16];
      sk_cone_pre(cdim, jar, cD, cmu, cimu, cfr, f, Hc);
      bool any = false;
#pragma unroll
      for (int q = 0; q < 16; q++) any = any || Hc[q] != 0;
      if (any) {
        // W rows: r < 3: [cm[r] ; cd[r]], r = 3: [cd[0] ; 0].  G = Hc W (4 x 6), K = W' G (lower triangle), F = W' f
        double G[4][6];
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma unroll
          for (int a = 0; a < 3; a++) {
            G[r][a] = Hc[4 * r] * cm[0][a] + Hc[4 * r + 1] * cm[1][a] + Hc[4 * r + 2] * cm[2][a] + Hc[4 * r + 3] * cd[0][a];
            G[r][3 + a] = Hc[4 * r] * cd[0][a] + Hc[4 * r + 1] * cd[1][a] + Hc[4 * r + 2] * cd[2][a];
          }
        }
        sk_lds_double* const Ks = Wb + SKW_SLOTK + slot * SKW_KST;
        sk_lds_double* const Fs = Wb + SKW_SLOTF + slot * 6;
#pragma unroll
        for (int a = 0; a < 6; a++) {
#pragma unroll
          for (int b = 0; b <= a; b++) {
            double v;
            if (a < 3) v = cm[0][a] * G[0][b] + cm[1][a] * G[1][b] + cm[2][a] * G[2][b] + cd[0][a] * G[3][b];
            else v = cd[0][a - 3] * G[0][b] + cd[1][a - 3] * G[1][b] + cd[2][a - 3] * G[2][b];
            (void)__hip_atomic_fetch_add(&Ks[a * (a + 1) / 2 + b], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
#pragma unroll
        for (int a = 0; a < 3; a++) {
          (void)__hip_atomic_fetch_add(&Fs[a], sgF * (f[0] * cm[0][a] + f[1] * cm[1][a] + f[2] * cm[2][a] + f[3] * cd[0][a]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          (void)__hip_atomic_fetch_add(&Fs[3 + a], sgF * (f[0] * cd[0][a] + f[1] * cd[1][a] + f[2] * cd[2][a]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
    }
    __syncthreads();
    SD_TOC(1);
    // aggregation: lane e < 21 sums entry e of the pair matrices into the tables the dof lanes read; lanes 21 .. 26: the net wrenches
    if (!fin && active) {
      if (hl < 21) {
        double v[16];
#pragma unroll
        for (int s = 0; s < 16; s++) v[s] = Wb[SKW_SLOTK + s * SKW_KST + hl];
        const double a0 = v[6] + v[9] + v[10], a1 = v[7] + v[11] + v[12], a2 = v[8] + v[13] + v[14];
        const double f0 = v[9] + v[11] + v[13], f1 = v[10] + v[12] + v[14];
        sk_lds_double* const A = Wb + SKW_AGGK + hl;
        A[0] = v[0] + v[3] + v[4] + a0; A[SKW_KST] = v[1] + v[3] + v[5] + a1; A[2 * SKW_KST] = v[2] + v[4] + v[5] + a2;
        A[3 * SKW_KST] = a0; A[4 * SKW_KST] = a1; A[5 * SKW_KST] = a2;
        A[6 * SKW_KST] = f0; A[7 * SKW_KST] = f1; A[8 * SKW_KST] = f0 + v[15]; A[9 * SKW_KST] = f1 + v[15];
        A[10 * SKW_KST] = a0 + a1 + a2;
      } else if (hl < 27) {
        const int c = hl - 21;
        double v[16];
#pragma unroll
        for (int s = 0; s < 16; s++) v[s] = Wb[SKW_SLOTF + 6 * s + c];
        const double n3 = v[6] + v[7] + v[8], n4 = v[9] + v[11] + v[13] - v[15], n5 = v[10] + v[12] + v[14] + v[15];
        sk_lds_double* const N = Wb + SKW_AGGN + c;
        N[0] = v[0] - v[6] - v[9] - v[10] - v[3] - v[4];
        N[6] = v[1] - v[7] - v[11] - v[12] + v[3] - v[5];
        N[12] = v[2] - v[8] - v[13] - v[14] + v[4] + v[5];
        N[18] = n3 + n4 + n5; N[24] = n4; N[30] = n5;
      }
    }
    __syncthreads();
    SD_TOC(2);
    // dof lanes: gradient entry and Hessian row
    double gl = mxa;
    limact = false;
    if (lsg != 0) { const double lj = lsg * xi - lar; if (lj < 0) { gl += lsg * lD * lj; limact = true; } }
    {
      const int cls = armrow ? (slide ? 4 + fs : 3) : bi;
      const sk_lds_double* const N = Wb + SKW_AGGN + 6 * cls;
      const double n0 = N[0], n1 = N[1], n2 = N[2], n3 = N[3], n4 = N[4], n5 = N[5];
      gl -= (si[0] * n0 + si[1] * n1 + si[2] * n2) + (si[3] * n3 + si[4] * n4 + si[5] * n5);
    }
    gi = (row && !fin) ? gl : 0.0;
    SD_TOC(4);
    return sk_half_max(fabs(gi), upper);
  };
  // the Hessian rows (only for a half that goes on: the pass that finds the gradient below the tolerance builds none)
  auto assemble = [&]() {
#pragma unroll
    for (int k = 0; k < SK_NV; k++) {
      double m = 0;
      if (k < SK_ARM0) m = (row && !armrow && k == i) ? mdiag : 0.0;
      else m = (armrow && k - SK_ARM0 <= ia) ? Ma[k - SK_ARM0] : 0.0;
      if (limact && k == i) m += lD;
      Hr[k] = m;
    }
#pragma unroll
    for (int cb = 0; cb <= SK_NB; cb++) {
      const bool mine = row && !fin && cb <= bi && ((cpl0 >> (4 * bi + cb)) & 1u);
      if (!__any(mine)) continue;
      int koff; double sgn = 1;
      if (cb < SK_NB) {
        if (bi < SK_NB) { koff = cb == bi ? SKW_AGGK + bi * SKW_KST : SKW_SLOTK + (3 + bi + cb - 1) * SKW_KST; sgn = cb == bi ? 1.0 : -1.0; }
        else if (!slide) { koff = SKW_AGGK + (3 + cb) * SKW_KST; sgn = -1.0; }
        else { koff = SKW_SLOTK + (9 + 2 * cb + fs) * SKW_KST; sgn = -1.0; }
      } else koff = slide ? SKW_AGGK + (6 + fs) * SKW_KST : SKW_AGGK + 10 * SKW_KST;
      if (!mine) { koff = SKW_AGGK; sgn = 0; }
      const sk_lds_double* const Kp = Wb + koff;
      double Kv[21], u[6];
#pragma unroll
      for (int q = 0; q < 21; q++) Kv[q] = Kp[q];
      const double sa[6] = {sgn * si[0], sgn * si[1], sgn * si[2], sgn * si[3], sgn * si[4], sgn * si[5]};
#pragma unroll
      for (int a = 0; a < 6; a++) {
        double s0 = 0, s1 = 0;      // two chains
#pragma unroll
        for (int b = 0; b < 3; b++) s0 += Kv[a >= b ? a * (a + 1) / 2 + b : b * (b + 1) / 2 + a] * sa[b];
#pragma unroll
        for (int b = 3; b < 6; b++) s1 += Kv[a >= b ? a * (a + 1) / 2 + b : b * (b + 1) / 2 + a] * sa[b];
        u[a] = s0 + s1;
      }
      const int j0 = cb < SK_NB ? 6 * cb : SK_ARM0, nj = cb < SK_NB ? 6 : NDOF;
#pragma unroll
      for (int jg = 0; jg < NDOF; jg += 3) {      // three columns at a time: their 18 words are read first, then used
        if (jg >= nj) continue;
        double sj[3][6];
#pragma unroll
        for (int c = 0; c < 3; c++) {
#pragma unroll
          for (int a = 0; a < 6; a++) sj[c][a] = Wb[SKW_SCOL + 6 * (j0 + jg + c) + a];
        }
#pragma unroll
        for (int c = 0; c < 3; c++) {
          const int j = j0 + jg + c;
          const double h = (u[0] * sj[c][0] + u[1] * sj[c][1] + u[2] * sj[c][2]) + (u[3] * sj[c][3] + u[4] * sj[c][4] + u[5] * sj[c][5]);
          if (mine && j <= i) Hr[j] += h;
        }
      }
    }
    // finger slides: dof 25 does not move finger 1 and vice versa, the finger <-> finger pair couples the two slides only
    if (slide && fs == 1) Hr[SK_ARM0 + NARM] = 0;
    if (has_ff) {
      const sk_lds_double* const Kp = Wb + SKW_SLOTK + 15 * SKW_KST;
      double uF[3];      // lower-right 3 x 3 block of FF times the slide axis
#pragma unroll
      for (int a = 0; a < 3; a++) uF[a] = Kp[sk_tri6(3 + a, 3)] * si[3] + Kp[sk_tri6(3 + a, 4)] * si[4] + Kp[sk_tri6(3 + a, 5)] * si[5];
      const sk_lds_double* const s25 = Wb + SKW_SCOL + 6 * (SK_ARM0 + NARM);
      const double hd = uF[0] * si[3] + uF[1] * si[4] + uF[2] * si[5], hx = uF[0] * s25[3] + uF[1] * s25[4] + uF[2] * s25[5];
      if (slide && !fin) {
        if (fs == 0) Hr[SK_ARM0 + NARM] += hd; else { Hr[SK_ARM0 + NARM + 1] += hd; Hr[SK_ARM0 + NARM] = -hx; }
      }
    }
    if (!fin) {      // the pair slots are clear again for the next pass (the dof lanes above were their last readers; the next additions come after a barrier)
#pragma unroll
      for (int q = 0; q < (16 * SKW_KST + 16 * 6 + 31) / 32; q++) { const int w = hl + 32 * q; if (w < 16 * SKW_KST + 16 * 6) Wb[SKW_SLOTK + w] = 0; }
    }
  };
  for (int it = 0; it < 60 && __any(!fin); it++) {
    const double gm = grad_pass();
    SD_TOC(8);
    if (lane == 0 && blockIdx.x == 0) { D3IL_SD_COUNT(7); }
    if (!fin && gm <= g_solver_tol.grad_tol) fin = true;      // converged
    if (!__any(!fin)) break;
    assemble();
    SD_TOC(5);
    // ---- Cholesky: lane i of a half holds row i
    // the forward substitution L y = -g rides along: the numerator of y_j is complete in lane j when column j is factorised and is broadcast together with the pivot
    double Lr[SK_NV], dinv = 1;
    double y = (row && !fin) ? -gi : 0.0;
    {
#pragma unroll
      for (int j = 0; j < SK_NV; j++) {
        double sum = Hr[j], s2 = 0;      // two accumulators: the dependent FMA chain is half as long
        constexpr int NBLK = SK_NB + 1;
        const int bj = j >= SK_ARM0 ? SK_NB : j / 6;
#pragma unroll
        for (int bk = 0; bk < NBLK; bk++) {
          if (bk > bj) continue;
          if (bk < bj && !((cpl >> (4 * bj + bk)) & 1u)) continue;      // row j of the factor has no entries in this column block (wave-uniform)
          const int k0 = bk == SK_NB ? SK_ARM0 : 6 * bk, k1 = bk == SK_NB ? SK_NV : 6 * bk + 6;
          double bc[NDOF];      // the broadcasts of a block are independent: issued back to back, one wait
#pragma unroll
          for (int k = k0; k < k1; k++) if (k < j) bc[k - k0] = sk_hbcast(Lr[k], j, upper);
#pragma unroll
          for (int k = k0; k < k1; k++) if (k < j) { if (k & 1) s2 -= Lr[k] * bc[k - k0]; else sum -= Lr[k] * bc[k - k0]; }
        }
        sum += s2;
        double sj = sk_hbcast_chain(sum, j, upper);
        const double yn = sk_hbcast(y, j, upper);
        if (!(sj > 0)) { if (!fin) okh = false; sj = 1; }
#if defined(D3IL_SK_PIVOT_2NEWTON)
        const double di = rsqrtd(sj), d = sj * di;      // v_rsq_f64 + two Newton steps instead of a square root and a division
#else
        const double di = sk_rsqrt1(sj), d = sj * di;
#endif
        Lr[j] = i == j ? d : (i > j ? sum * di : 0.0);
        if (i == j) dinv = di;
        const double yj = yn * di;
        y = i == j ? yj : (i > j ? y - Lr[j] * yj : y);
      }
    }
    if (!okh) fin = true;      // non-positive pivot: this half gives up
    SD_TOC(9);
    // ---- p = -H^-1 g: backward substitution with the columns of the factor (handed over through Hs)
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SK_NV; k++) if (row && !fin && k <= i) Hs[tri(i, k)] = Lr[k];
    __syncthreads();
    {
      double Lc[SK_NV];
#pragma unroll
      for (int k = 0; k < SK_NV; k++) { const bool in = row && k > i; const double v = Hs[(in ? tri(k, i) : 0)]; Lc[k] = in ? v : 0.0; }
#pragma unroll
      for (int j = SK_NV - 1; j >= 0; j--) {
        const double xj = sk_hbcast(y * dinv, j, upper);
        y = i == j ? xj : (i < j ? y - Lc[j] * xj : y);
      }
    }
    const double pi = (row && !fin) ? y : 0.0;
    if (row && !fin) Ps[i] = pi;
    __syncthreads();
    SD_TOC(10);
    // ---- line search: phi'(alpha) = p' M (x - a0) + alpha p' M p - sum f(jar + alpha Jp) . Jp, safeguarded Newton on alpha
    twists(Ps);
    const double Mp = fin ? 0.0 : m_times(Ps, nullptr);
    const double gTp = sk_half_sum(gi * pi, upper), pMp = sk_half_sum(pi * Mp, upper), pMa = sk_half_sum(pi * mxa, upper);
    __syncthreads();
    double jp[4] = {0, 0, 0, 0};
    if (con && !fin) wdot(jp);
    const double ljp = lsg * pi, ljar = lsg * xi - lar;
    SD_TOC(11);
    double alpha = 1, lo = 0, hi = -1, best = 1, wprev = 1e300;
    bool lsdone = fin;
    for (int ls = 0; ls < 50 && __any(!lsdone); ls++) {
      double d1c = 0, d2c = 0;
      if (lsg != 0) { const double lj = ljar + alpha * ljp; if (lj < 0) { d1c += lD * lj * ljp; d2c += lD * ljp * ljp; } }
      if (con && !lsdone) {
        double jt[4], f[4], Hc[16];
#pragma unroll
        for (int r = 0; r < 4; r++) jt[r] = jar[r] + alpha * jp[r];
        sk_cone_pre(cdim, jt, cD, cmu, cimu, cfr, f, Hc);
#pragma unroll
        for (int r = 0; r < 4; r++) { d1c -= f[r] * jp[r];
#pragma unroll
          for (int q = 0; q < 4; q++) d2c += jp[r] * Hc[4 * r + q] * jp[q]; }
      }
      const double d1 = pMa + alpha * pMp + sk_half_sum(d1c, upper), d2 = pMp + sk_half_sum(d2c, upper);
      if (lsdone) continue;
      best = alpha;
      if (ls == 0 && d1 <= g_solver_tol.ls_full * fabs(gTp)) { lsdone = true; continue; }
      if (fabs(d1) <= g_solver_tol.ls_c2 * fabs(gTp) || fabs(d1) <= g_solver_tol.ls_rel * d2 * alpha || fabs(d1) < 1e-14 * fmax(1.0, fabs(pMa))) { lsdone = true; continue; }
      if (d1 < 0) lo = alpha; else hi = alpha;
      double na = alpha - d1 / d2;
      if (hi >= 0) {
        const double wbr = hi - lo;
        const bool slow = wbr > 0.5 * wprev;
        wprev = wbr;
        if (slow || !(na > lo && na < hi)) na = 0.5 * (lo + hi);
      } else if (na <= lo) na = 2 * lo + 1;
      if (na == alpha) { lsdone = true; continue; }
      alpha = na;
    }
    SD_TOC(12);
    const double dx = fin ? 0.0 : best * pi;
    xi += dx;
    if (!fin) {
      mxa += best * Mp;
#pragma unroll
      for (int r = 0; r < 4; r++) jar[r] += best * jp[r];
    }
    if (row && !fin) t[ST_X + i] = xi;
    const double smax = sk_half_max(fabs(dx), upper), xmax = sk_half_max(fabs(xi), upper);
    if (!fin && (smax <= 1e-12 * (1 + xmax) || (best == 1.0 && smax <= g_solver_tol.step_rel * (1 + xmax)))) fin = true;      // converged
    __syncthreads();
  }
#undef SD_TOC
  if (active && !fin) okh = false;      // iteration cap
  const unsigned long long bad = __ballot(active && !okh);
  return (unsigned)((bad & 0xFFFFFFFFull) != 0 ? 1u : 0u) | (unsigned)((bad >> 32) != 0 ? 2u : 0u);
}
#endif

// ------------------------------------------------------------------------------------------------ sub-step
struct StackState { EnvState arm; BoxState box[SK_NB]; double warm[SK_NV]; };

D3IL_HD void sk_add_contact(const StackConsts& kc_, const StackScratch sc, int& ncon, unsigned& flags, const double* rec7, double nsign, int bodyA, int bodyB, int set) {
  if (ncon >= SK_MAXCON) { flags |= SKF_CON_OVERFLOW; return; }
  const int base = ncon * SREC;
  double n[3] = {nsign * rec7[4], nsign * rec7[5], nsign * rec7[6]}, t1[3], t2[3];
  make_frame(n, t1, t2);
  SG(base + 0) = rec7[1]; SG(base + 1) = rec7[2]; SG(base + 2) = rec7[3];
  for (int k = 0; k < 3; k++) { SG(base + 3 + k) = n[k]; SG(base + 6 + k) = t1[k]; SG(base + 9 + k) = t2[k]; }
  SG(base + 12) = rec7[0]; SG(base + 13) = (double)bodyA; SG(base + 14) = (double)bodyB; SG(base + 15) = (double)set;
  ncon++;
}
// One physics sub-step (mj_step) with the torques of this sub-step's control law, in three parts: stack_substep_pre (kinematics,
// smooth accelerations, collision, limit rows, start point of the solver), the constraint solve (sk_solve on one lane, or
// sk_solve_coop by the whole wave), stack_substep_post (mj_Euler).  WARM_LDS: the warm start is the x vector left in the t area by
// the previous sub-step (device step kernel) instead of ss.warm.
template <int V = SKV_STACKING, class C>
D3IL_HD void stack_pre_kin(const C& c0, const StackConsts& kc_, StackState& ss, const StackScratch sc, const double* tau, const double* ffing, const double* trig = nullptr) {
  D3IL_STACK_CONSTS(kc_, kc);
  D3IL_REFRESH(c0, c);
  EnvState& st = ss.arm;
  const double h = c.timestep;
  SK_TIC;
  // ---- arm forward pass
  DynOut dyn;
  dynamics(c0, st.q, st.v, dyn, trig);
  double fs[NDOF];
  for (int k = 0; k < NARM; k++) fs[k] = clampd(tau[k] + st.bias[k], c.force_lo[k], c.force_hi[k]) - dyn.bias[k];
  for (int k = 0; k < NFING; k++) fs[NARM + k] = clampd(ffing[k], c.force_lo[NARM + k], c.force_hi[NARM + k]) - dyn.bias[NARM + k] - c.f_damping[k] * st.v[NARM + k];
  for (int k = 0; k < NARM; k++) st.bias[k] = dyn.bias[k];
  {
    double t[3]; mulE(dyn.R7, c.tcp7, t);
    st.tcp[0] = dyn.p7[0] + t[0]; st.tcp[1] = dyn.p7[1] + t[1]; st.tcp[2] = dyn.p7[2] + t[2];
  }
  for (int k = 0; k < 45; k++) SL(ST_M + k) = dyn.M[k];
  {   // smooth acceleration of the arm
    double L[45], d[NDOF], id[NDOF], a0[NDOF];
    if (!ldl9(dyn.M, L, d, id)) st.flags |= F_SOLVER_FAIL;
    for (int k = 0; k < NDOF; k++) a0[k] = fs[k];
    ldl9_solve(L, id, a0);
    for (int k = 0; k < NDOF; k++) { SL(ST_A0 + SK_ARM0 + k) = a0[k]; SL(ST_VEL + SK_ARM0 + k) = st.v[k]; }
  }
  {   // world joint axes / origins, finger slide axes, finger geom poses
    double R7[9], p7[3], ax[NARM][3], og[NARM][3];
    world_chain(c0, dyn.sn, dyn.cs, R7, p7, ax, og);
    for (int k = 0; k < NARM; k++) for (int i = 0; i < 3; i++) { SL(ST_Z + 3 * k + i) = ax[k][i]; SL(ST_O + 3 * k + i) = og[k][i]; }
    if constexpr (V == SKV_ALIGNING) {      // rod cylinder: centre and axis in the world
      double t3[3];
      mulE(R7, kc.rod_c7, t3);
      for (int i = 0; i < 3; i++) SL(ST_TIPR + SV_ROD + i) = p7[i] + t3[i];
      mulE(R7, kc.rod_u7, t3);
      for (int i = 0; i < 3; i++) SL(ST_TIPR + SV_ROD + 3 + i) = t3[i];
    }
    for (int f = 0; f < NFING && V == SKV_STACKING; f++) {
      double axw[3]; mulE(R7, c.f_axis[f], axw);
      for (int i = 0; i < 3; i++) SL(ST_FAX + 3 * f + i) = axw[i];
      const double qf = st.q[NARM + f];
      for (int g = 0; g < 2; g++) {
        const double* Rl = g ? kc.hull_R[f] : kc.tip_R[f];
        const double* pl = g ? kc.hull_p[f] : kc.tip_p[f];
        const int oR = (g ? ST_HULR : ST_TIPR) + 9 * f, oP = (g ? ST_HULP : ST_TIPP) + 3 * f;
        double pm[3] = {pl[0] + c.f_axis[f][0] * qf, pl[1] + c.f_axis[f][1] * qf, pl[2] + c.f_axis[f][2] * qf}, pw[3];
        mulE(R7, pm, pw);
        for (int i = 0; i < 3; i++) SL(oP + i) = p7[i] + pw[i];
        for (int r = 0; r < 3; r++) for (int cc = 0; cc < 3; cc++) SL(oR + 3 * r + cc) = R7[3 * r] * Rl[cc] + R7[3 * r + 1] * Rl[3 + cc] + R7[3 * r + 2] * Rl[6 + cc];
      }
    }
  }
  // ---- boxes: rotation matrices, smooth acceleration (gravity, gyroscopic term of the body-frame angular dofs)
  for (int b = 0; b < SK_NB; b++) {
    double R[9], qn[4];     // mj_kinematics works on the normalised quaternion (a context may be off by float32 round-off)
    { const double* q = ss.box[b].quat; const double nn = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]); for (int k = 0; k < 4; k++) qn[k] = q[k] / nn; }
    quat2mat(qn, R);
    for (int k = 0; k < 9; k++) SL(ST_BR + 9 * b + k) = R[k];
    for (int k = 0; k < 3; k++) SL(ST_BP + 3 * b + k) = ss.box[b].pos[k];
    if (V == SKV_ALIGNING && b == 0) {
      // centre-of-mass coordinates of the compound body: p_c = p_o + R c, v_c = v_o + R (w x c) (w: body-frame angular velocity, MuJoCo's free-joint
      // convention); centripetal term w_w x (w_w x R c) = R (w x (w x c)) for the contact rows and the conversion back
      const double* wl = ss.box[0].vel + 3;
      double wc[3], wwc[3], t3[3];
      cross3(wl, kc.al_c, wc); cross3(wl, wc, wwc);
      mulE(R, kc.al_c, t3); for (int k = 0; k < 3; k++) SL(ST_BP + k) = ss.box[0].pos[k] + t3[k];
      mulE(R, wc, t3); for (int k = 0; k < 3; k++) ss.box[0].vel[k] += t3[k];
      mulE(R, wwc, t3); for (int k = 0; k < 3; k++) SL(ST_TIPR + SV_CEN + k) = t3[k];
    }
    const double* I = kc.box_inertia[b];
    const double* w = ss.box[b].vel + 3;
    const double Iw[3] = {I[0] * w[0], I[1] * w[1], I[2] * w[2]};
    double gy[3]; cross3(w, Iw, gy);
    const bool inert = b >= kc.nb;      // a block this variant does not use: parked far away, no gravity, never touched by a constraint
    for (int k = 0; k < 3; k++) { SL(ST_A0 + 6 * b + k) = inert ? 0.0 : c.gravity[k]; SL(ST_A0 + 6 * b + 3 + k) = inert ? 0.0 : -gy[k] / I[k]; }
    for (int k = 0; k < 6; k++) SL(ST_VEL + 6 * b + k) = ss.box[b].vel[k];
    if (!inert && (ss.box[b].pos[0] < kc.ws_lo[0] || ss.box[b].pos[0] > kc.ws_hi[0] || ss.box[b].pos[1] < kc.ws_lo[1] || ss.box[b].pos[1] > kc.ws_hi[1])) st.flags |= SKF_OFF_TABLE;
  }
  SL(ST_AUX) = st.q[NARM] + st.q[NARM + 1];
  SK_TOC(0);
}
// World pose of the hand geom in this sub-step, from the tables the kinematics phase wrote: link 7's frame is recovered from the left
// finger tip's (tip = R7 tip_R[0], R7 (tip_p[0] + f_axis[0] qf) + p7), then the hand geom's constant pose in the link-7 frame is applied.
// TS: reads the t area (LDS view or plain memory)
template <class TS>
D3IL_HD void sk_hand_pose(const StackConsts& kc_, const TS t, const double qf /* position of the left finger joint */, double* Rh, double* ph) {
  D3IL_STACK_CONSTS(kc_, kc);
  double TR[9], R7[9], pl[3], pw[3];
  for (int k = 0; k < 9; k++) TR[k] = t[ST_TIPR + k];
  for (int r = 0; r < 3; r++) for (int cc = 0; cc < 3; cc++) R7[3 * r + cc] = TR[3 * r] * kc.tip_R[0][3 * cc] + TR[3 * r + 1] * kc.tip_R[0][3 * cc + 1] + TR[3 * r + 2] * kc.tip_R[0][3 * cc + 2];
  for (int k = 0; k < 3; k++) pl[k] = kc.tip_p[0][k] + kc.f_axis0[k] * qf;
  mulE(R7, pl, pw);
  double p7[3] = {t[ST_TIPP] - pw[0], t[ST_TIPP + 1] - pw[1], t[ST_TIPP + 2] - pw[2]};
  for (int r = 0; r < 3; r++) for (int cc = 0; cc < 3; cc++) Rh[3 * r + cc] = R7[3 * r] * kc.hand_R[cc] + R7[3 * r + 1] * kc.hand_R[3 + cc] + R7[3 * r + 2] * kc.hand_R[6 + cc];
  mulE(R7, kc.hand_p, pw);
  for (int k = 0; k < 3; k++) ph[k] = p7[k] + pw[k];
}
// exact cull of a box <-> hand pair: the box against the bounding box of the hand hull (in the hand geom frame) grown by the margin
D3IL_HD bool sk_hand_near(const StackConsts& kc_, const double* Rh, const double* ph, const double* Rb, const double* pb, const double* hb, double margin) {
  D3IL_STACK_CONSTS(kc_, kc);
  const double dw[3] = {pb[0] - ph[0], pb[1] - ph[1], pb[2] - ph[2]};
  for (int i = 0; i < 3; i++) {
    const double ci = Rh[i] * dw[0] + Rh[3 + i] * dw[1] + Rh[6 + i] * dw[2];
    double ei = 0;
    for (int j = 0; j < 3; j++) ei += fabs(Rh[i] * Rb[j] + Rh[3 + i] * Rb[3 + j] + Rh[6 + i] * Rb[6 + j]) * hb[j];
    if (ci - ei > kc.hand_hi[i] + margin || ci + ei < kc.hand_lo[i] - margin) return false;
  }
  return true;
}
// collision on one lane, in the model's geom order: static < boxes < left hull < left tip < right hull < right tip
D3IL_NOINLINE inline int stack_pre_collide(const StackConsts& kc_, StackState& ss, const StackScratch sc) {
  D3IL_STACK_CONSTS(kc_, kc);
  EnvState& st = ss.arm;
  SK_TIC;
  int ncon = 0;
  double rec[8][7];
  auto boxR = [&](int b, double* R) { for (int k = 0; k < 9; k++) R[k] = SL(ST_BR + 9 * b + k); };
  for (int s = 0; s < kc.ns; s++)
    for (int b = 0; b < SK_NB; b++) {
      double R[9]; boxR(b, R);
      // sphere against the static box (exact distance of the centre from the box)
      double d[3] = {ss.box[b].pos[0] - kc.st_c[s][0], ss.box[b].pos[1] - kc.st_c[s][1], ss.box[b].pos[2] - kc.st_c[s][2]}, ex = 0;
      for (int i = 0; i < 3; i++) { double loc = kc.st_R[s][i] * d[0] + kc.st_R[s][3 + i] * d[1] + kc.st_R[s][6 + i] * d[2]; double o = fabs(loc) - kc.st_h[s][i]; if (o > 0) ex += o * o; }
      const double rc = kc.box_r[b] + kc.set[SKS_STATIC + s].margin;
      if (ex > rc * rc) continue;
      const int n = box_box(kc.st_c[s], kc.st_R[s], kc.st_h[s], ss.box[b].pos, R, kc.box_half[b], kc.set[SKS_STATIC + s].margin, rec, 8);
      for (int i = 0; i < n; i++) sk_add_contact(kc, sc, ncon, st.flags, rec[i], 1.0, SKB_STATIC, b, SKS_STATIC + s);
    }
  for (int b1 = 0; b1 < SK_NB; b1++)
    for (int b2 = b1 + 1; b2 < SK_NB; b2++) {
      double d[3] = {ss.box[b2].pos[0] - ss.box[b1].pos[0], ss.box[b2].pos[1] - ss.box[b1].pos[1], ss.box[b2].pos[2] - ss.box[b1].pos[2]};
      const double rc = kc.box_r[b1] + kc.box_r[b2] + kc.set[SKS_BOXBOX].margin;
      if (dot3(d, d) > rc * rc) continue;
      double R1[9], R2[9]; boxR(b1, R1); boxR(b2, R2);
      const int n = box_box(ss.box[b1].pos, R1, kc.box_half[b1], ss.box[b2].pos, R2, kc.box_half[b2], kc.set[SKS_BOXBOX].margin, rec, 8);
      for (int i = 0; i < n; i++) sk_add_contact(kc, sc, ncon, st.flags, rec[i], 1.0, b1, b2, SKS_BOXBOX);
    }
  SK_TOC(1);
  const double r_tip = kc.tip_r, r_hull = kc.hull_r;
  double fR[NFING][2][9], fP[NFING][2][3], hullC[NFING][3];
  for (int f = 0; f < NFING; f++) {
    for (int k = 0; k < 9; k++) { fR[f][0][k] = SL(ST_HULR + 9 * f + k); fR[f][1][k] = SL(ST_TIPR + 9 * f + k); }
    for (int k = 0; k < 3; k++) { fP[f][0][k] = SL(ST_HULP + 3 * f + k); fP[f][1][k] = SL(ST_TIPP + 3 * f + k); }
    for (int k = 0; k < 3; k++) hullC[f][k] = fR[f][0][3 * k] * kc.hull_center[0] + fR[f][0][3 * k + 1] * kc.hull_center[1] + fR[f][0][3 * k + 2] * kc.hull_center[2] + fP[f][0][k];
  }
  {   // box <-> hand hull (panda_hand:geom2): exact bounding-box cull, then MPR on the 773-vertex hull
    double Rh[9], ph[3];
    sk_hand_pose(kc, sc.t, st.q[NARM], Rh, ph);
    for (int b = 0; b < SK_NB; b++) {
      double R[9]; boxR(b, R);
      if (!sk_hand_near(kc, Rh, ph, R, ss.box[b].pos, kc.box_half[b], kc.set[SKS_BOXHAND].margin)) continue;
      SkShape A{R, ss.box[b].pos, kc.box_half[b], 0}, B{Rh, ph, nullptr, 2};
      double r7[7];
      if (sk_mpr(kc, A, B, kc.set[SKS_BOXHAND].margin, r7)) sk_add_contact(kc, sc, ncon, st.flags, r7, 1.0, b, SKB_HAND, SKS_BOXHAND);
    }
  }
  for (int b = 0; b < SK_NB; b++) {
    double R[9]; boxR(b, R);
    const double rb = kc.box_r[b];
    for (int f = 0; f < NFING; f++) {
      {   // box <-> finger hull (geom order: box first)
        double d[3] = {hullC[f][0] - ss.box[b].pos[0], hullC[f][1] - ss.box[b].pos[1], hullC[f][2] - ss.box[b].pos[2]};
        const double rc = rb + r_hull + kc.set[SKS_BOXHULL].margin;
        if (dot3(d, d) <= rc * rc) {
          SkShape A{R, ss.box[b].pos, kc.box_half[b], 0}, B{fR[f][0], fP[f][0], nullptr, 1};
          double r7[7];
          if (sk_mpr(kc, A, B, kc.set[SKS_BOXHULL].margin, r7)) sk_add_contact(kc, sc, ncon, st.flags, r7, 1.0, b, SKB_FINGER + f, SKS_BOXHULL);
        }
      }
      {   // box <-> finger-tip box
        double d[3] = {fP[f][1][0] - ss.box[b].pos[0], fP[f][1][1] - ss.box[b].pos[1], fP[f][1][2] - ss.box[b].pos[2]};
        const double rc = rb + r_tip + kc.set[SKS_BOXTIP].margin;
        if (dot3(d, d) <= rc * rc) {
          const int n = box_box(ss.box[b].pos, R, kc.box_half[b], fP[f][1], fR[f][1], kc.tip_half, kc.set[SKS_BOXTIP].margin, rec, 8);
          for (int i = 0; i < n; i++) sk_add_contact(kc, sc, ncon, st.flags, rec[i], 1.0, b, SKB_TIP + f, SKS_BOXTIP);
        }
      }
    }
  }
  if (SL(ST_AUX) < 0.004) {     // finger <-> finger: only a (nearly) closed gripper (the gaps are q1 + q2 - 1 mm or less)
    double r7[7];
    {
      SkShape A{fR[0][0], fP[0][0], nullptr, 1}, B{fR[1][0], fP[1][0], nullptr, 1};
      if (sk_mpr(kc, A, B, kc.set[SKS_HULLHULL].margin, r7)) sk_add_contact(kc, sc, ncon, st.flags, r7, 1.0, SKB_FINGER, SKB_FINGER + 1, SKS_HULLHULL);
    }
    {
      SkShape A{fR[0][0], fP[0][0], nullptr, 1}, B{fR[1][1], fP[1][1], kc.tip_half, 0};
      if (sk_mpr(kc, A, B, kc.set[SKS_HULLTIP].margin, r7)) sk_add_contact(kc, sc, ncon, st.flags, r7, 1.0, SKB_FINGER, SKB_TIP + 1, SKS_HULLTIP);
    }
    {
      SkShape A{fR[0][1], fP[0][1], kc.tip_half, 0}, B{fR[1][0], fP[1][0], nullptr, 1};
      if (sk_mpr(kc, A, B, kc.set[SKS_HULLTIP].margin, r7)) sk_add_contact(kc, sc, ncon, st.flags, r7, 1.0, SKB_TIP, SKB_FINGER + 1, SKS_HULLTIP);
    }
    const int n = box_box(fP[0][1], fR[0][1], kc.tip_half, fP[1][1], fR[1][1], kc.tip_half, kc.set[SKS_TIPTIP].margin, rec, 8);
    for (int i = 0; i < n; i++) sk_add_contact(kc, sc, ncon, st.flags, rec[i], 1.0, SKB_TIP, SKB_TIP + 1, SKS_TIPTIP);
  }
  SK_TOC(2);
  return ncon;
}
template <bool WARM_LDS, class C>
D3IL_HD void stack_pre_finish(const C& c0, const StackConsts& kc_, StackState& ss, const StackScratch sc, const int ncon, unsigned has, bool& any_lim_out) {
  D3IL_REFRESH(c0, c);
  EnvState& st = ss.arm;
  SK_TIC;
  // ---- joint-limit rows (mj_instantiateLimit) of the 9 arm dofs
  bool any_lim = false;
  for (int k = 0; k < NDOF; k++) {
    const double dlo = st.q[k] - c.jnt_range[k][0], dhi = c.jnt_range[k][1] - st.q[k];
    double sign = 0, dist = 0;
    if (dlo < c.lim_margin[k]) { sign = 1; dist = dlo; } else if (dhi < c.lim_margin[k]) { sign = -1; dist = dhi; }
    double D = 0, ar = 0;
    if (sign != 0) {
      const double imp = impedance(c.lim_solimp[k], dist - c.lim_margin[k]);
      D = 1 / fmax(1e-15, (1 - imp) / imp * c.dof_invweight0[k]);
      ar = -c.lim_B[k] * (sign * st.v[k]) - c.lim_K[k] * imp * (dist - c.lim_margin[k]);
      any_lim = true;
    }
    SL(ST_LIM + 3 * k) = sign; SL(ST_LIM + 3 * k + 1) = D; SL(ST_LIM + 3 * k + 2) = ar;
  }
  SK_TOC(3);
  // start point of the solver: the previous sub-step's accelerations (MuJoCo's qacc_warmstart) for the blocks that carry constraints,
  // the smooth accelerations otherwise (those blocks are not moved by the solver)
  const bool warm = (st.flags & SKF_WARM_VALID) != 0;
  if (any_lim) has |= 1u << SK_NB;
  for (int b = 0; b <= SK_NB; b++) {
    const bool keep = warm && ((has >> b) & 1u);
    for (int i = sk_blk0(b); i < sk_blk0(b) + sk_blkn(b); i++) {
      if (!keep) SL(ST_X + i) = SL(ST_A0 + i);
      else if (!WARM_LDS) SL(ST_X + i) = ss.warm[i];
    }
  }
#if !defined(D3IL_DEVICE_STATS)
  if (!WARM_LDS)
#endif
  { SG(SG_DIAG) = 0; SG(SG_DIAG + 1) = 0; SG(SG_DIAG + 2) = 1; SG(SG_DIAG + 3) = (double)ncon; }
  any_lim_out = any_lim;
  (void)kc_;
}
// blocks that carry contacts (bit b: box b, bit SK_NB: the arm), from the records
D3IL_HD unsigned sk_has_mask(const StackScratch sc, int ncon) {
  unsigned has = 0;
  for (int ci = 0; ci < ncon; ci++) { const int ba = sk_blk_of((int)SG(ci * SREC + 13)); has |= 1u << sk_blk_of((int)SG(ci * SREC + 14)); if (ba >= 0) has |= 1u << ba; }
  return has;
}
template <bool WARM_LDS, class C>
D3IL_HD void stack_substep_pre(const C& c0, const StackConsts& kc_, StackState& ss, const StackScratch sc, const double* tau, const double* ffing,
                               int& ncon_out, bool& any_lim_out) {
  stack_pre_kin(c0, kc_, ss, sc, tau, ffing);
  const int ncon = stack_pre_collide(kc_, ss, sc);
  stack_pre_finish<WARM_LDS>(c0, kc_, ss, sc, ncon, sk_has_mask(sc, ncon), any_lim_out);
  ncon_out = ncon;
}
#if defined(__HIPCC__)
// Support point of one shape by a group of EIGHT lanes that hold the same shape / direction: lane s of the group evaluates the hull
// vertices s, s + 8, ...; the maximum and the lowest index within 1e-10 of it come from reductions over the group (xor shuffles with
// masks 4, 2, 1 stay inside it) - the vertex sk_support1 finds with its two passes over the table.
constexpr int SKG = 8, SKG_NV = (SK_MAXHV + SKG - 1) / SKG;
// reductions over a group of eight lanes (all active together): quad permutes + half-row mirror as DPP moves - three VALU stages instead of
// three LDS-crossbar round trips (ds_bpermute) per reduction; -DD3IL_SK_SHFL_REDUCE keeps the shuffle version for comparison
__device__ __forceinline__ int sk_dpp_movi(int v, const int ctrl_sel) {
  switch (ctrl_sel) {
    case 0: return __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false);
    case 1: return __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false);
    default: return __builtin_amdgcn_update_dpp(v, v, 0x141, 0xF, 0xF, false);
  }
}
__device__ __forceinline__ double sk_group_max(double v) {
  SK_CONVERGE();
#if defined(D3IL_SK_SHFL_REDUCE)
#pragma unroll
  for (int m = SKG / 2; m >= 1; m >>= 1) v = fmax(v, __shfl_xor(v, m));
#else
  v = fmax(v, sk_dpp_mov(v, 0)); v = fmax(v, sk_dpp_mov(v, 1)); v = fmax(v, sk_dpp_mov(v, 2));
#endif
  SK_CONVERGE();
  return v;
}
__device__ __forceinline__ int sk_group_min(int v) {
  SK_CONVERGE();
#if defined(D3IL_SK_SHFL_REDUCE)
#pragma unroll
  for (int m = SKG / 2; m >= 1; m >>= 1) { const int o = __shfl_xor(v, m); v = o < v ? o : v; }
#else
#pragma unroll
  for (int st = 0; st < 3; st++) { const int o = sk_dpp_movi(v, st); v = o < v ? o : v; }
#endif
  SK_CONVERGE();
  return v;
}
__device__ __forceinline__ void sk_support1_group(const StackConsts& kc_, const SkShape& s, const double* dir, double margin, double* out, const int sub) {
  D3IL_STACK_CONSTS(kc_, kc);
  const double* R = s.R;
  double dl[3] = {R[0] * dir[0] + R[3] * dir[1] + R[6] * dir[2], R[1] * dir[0] + R[4] * dir[1] + R[7] * dir[2], R[2] * dir[0] + R[5] * dir[1] + R[8] * dir[2]};
  double loc[3];
  if (s.hull) {
    const int nv = kc.hull_nv;
    double d[SKG_NV], bd = -1e300;
#pragma unroll
    for (int m = 0; m < SKG_NV; m++) {
      const int v = sub + SKG * m;
      d[m] = v < nv ? kc.hull_v[v][0] * dl[0] + kc.hull_v[v][1] * dl[1] + kc.hull_v[v][2] * dl[2] : -1e300;
      bd = fmax(bd, d[m]);
    }
    bd = sk_group_max(bd);
    const double thr = bd - 1e-10;
    int best = 1 << 20;
#pragma unroll
    for (int m = SKG_NV - 1; m >= 0; m--) if (d[m] >= thr) best = sub + SKG * m;
    best = sk_group_min(best);
    loc[0] = kc.hull_v[best][0]; loc[1] = kc.hull_v[best][1]; loc[2] = kc.hull_v[best][2];
  } else {
#pragma unroll
    for (int k = 0; k < 3; k++) loc[k] = dl[k] >= -1e-10 ? s.half[k] : -s.half[k];
  }
#pragma unroll
  for (int k = 0; k < 3; k++) out[k] = R[3 * k] * loc[0] + R[3 * k + 1] * loc[1] + R[3 * k + 2] * loc[2] + s.p[k] + 0.5 * margin * dir[k];
}
// The same with the lane's share of the finger-hull vertices (sub, sub + 8, ...) held in registers for the whole MPR job: the table is read once
// per job instead of once per support evaluation (an MPR job evaluates the support function ~100 times)
__device__ __forceinline__ void sk_support1_group_pre(const StackConsts& kc_, const SkShape& s, const double (*hv)[3], const double* dir, double margin, double* out, const int sub) {
  D3IL_STACK_CONSTS(kc_, kc);
  const double* R = s.R;
  double dl[3] = {R[0] * dir[0] + R[3] * dir[1] + R[6] * dir[2], R[1] * dir[0] + R[4] * dir[1] + R[7] * dir[2], R[2] * dir[0] + R[5] * dir[1] + R[8] * dir[2]};
  double loc[3];
  if (s.hull) {
    const int nv = kc.hull_nv;
    double d[SKG_NV], bd = -1e300;
#pragma unroll
    for (int m = 0; m < SKG_NV; m++) {
      const int v = sub + SKG * m;
      d[m] = v < nv ? hv[m][0] * dl[0] + hv[m][1] * dl[1] + hv[m][2] * dl[2] : -1e300;
      bd = fmax(bd, d[m]);
    }
    bd = sk_group_max(bd);
    const double thr = bd - 1e-10;
    int best = 1 << 20, bm = 0;
#pragma unroll
    for (int m = SKG_NV - 1; m >= 0; m--) if (d[m] >= thr) { best = sub + SKG * m; bm = m; }
    const int gbest = sk_group_min(best);
    // the winning vertex sits in the registers of lane (gbest % 8) of the group, slot gbest / 8: fetched with three shuffles
    double mine[3] = {0, 0, 0};
#pragma unroll
    for (int m = 0; m < SKG_NV; m++) if (m == bm) { mine[0] = hv[m][0]; mine[1] = hv[m][1]; mine[2] = hv[m][2]; }
    const int src = (threadIdx.x & ~(SKG - 1)) | (gbest & (SKG - 1));
    // Fenced on both sides (SK_CONVERGE above): this is the place where the -DD3IL_SK_PRELOAD_RAW build - no fence anywhere - goes wrong; rounds
    // 2 - 3 shipped an opaque register move (asm volatile "+v") here, which worked because it is a side effect, not because it hides the values.
#if defined(D3IL_SK_EXP) && D3IL_SK_EXP == 1      // the experiments of DESIGN section 18.2 (on the fence-less build): what about that move mattered?
    asm("" : "+v"(mine[0]), "+v"(mine[1]), "+v"(mine[2]));            // opaque values, no side effect: still wrong
#elif defined(D3IL_SK_EXP) && D3IL_SK_EXP == 2
    asm volatile("" ::: "memory");                                     // a side effect, values transparent: right
#elif defined(D3IL_SK_EXP) && D3IL_SK_EXP == 3
    __builtin_amdgcn_wave_barrier();                                   // a convergent no-op: right
#elif defined(D3IL_SK_EXP) && D3IL_SK_EXP == 4
    asm volatile("" : "+v"(mine[0]));                                  // one of the three values through a volatile asm: right
#endif
    SK_CONVERGE();
#pragma unroll
    for (int k = 0; k < 3; k++) loc[k] = __shfl(mine[k], src);
    SK_CONVERGE();
  } else {
#pragma unroll
    for (int k = 0; k < 3; k++) loc[k] = dl[k] >= -1e-10 ? s.half[k] : -s.half[k];
  }
#pragma unroll
  for (int k = 0; k < 3; k++) out[k] = R[3 * k] * loc[0] + R[3 * k + 1] * loc[1] + R[3 * k + 2] * loc[2] + s.p[k] + 0.5 * margin * dir[k];
}
// shape of an MPR job with its pose in the LDS tables (re-read by every support evaluation: nothing of it stays live across the portal iterations)
struct SkShapeL { const sk_lds_double* R; const sk_lds_double* p; double half[3]; int hull; };
__device__ __forceinline__ void sk_support1_group_l(const StackConsts& kc_, const SkShapeL& s, const double* dir, double margin, double* out, const int sub) {
  double R[9], p[3];
#pragma unroll
  for (int k = 0; k < 9; k++) R[k] = s.R[k];
#pragma unroll
  for (int k = 0; k < 3; k++) p[k] = s.p[k];
  const SkShape sh{R, p, s.half, s.hull};
  sk_support1_group(kc_, sh, dir, margin, out, sub);
}
__device__ __forceinline__ void sk_support1_group_lp(const StackConsts& kc_, const SkShapeL& s, const double (*hv)[3], const double* dir, double margin, double* out, const int sub) {
  double R[9], p[3];
#pragma unroll
  for (int k = 0; k < 9; k++) R[k] = s.R[k];
#pragma unroll
  for (int k = 0; k < 3; k++) p[k] = s.p[k];
  const SkShape sh{R, p, s.half, s.hull};
  sk_support1_group_pre(kc_, sh, hv, dir, margin, out, sub);
}
// Collision of the workgroup's environments with one lane per (environment, pair group), lane = group * SK_LANES + environment:
//   group 0 .. 2  : box b against the static boxes          3 .. 5 : the box pairs (0, 1) (0, 2) (1, 2)
//   group 6 .. 11 : box b against finger f: tip, hull (MPR)    12  : finger <-> finger (nearly closed gripper)
//   group 13 .. 15: box b against the hand hull (773 vertices: MPR by the WHOLE wave, one job at a time, after an exact bounding-box cull)
// Jobs: every lane has at most one pair test per round and all lanes run the test of a round at ONE call site per kind (box-box, MPR),
// so that lanes of different groups do not serialise through separately inlined copies of the same routine.
//   round   group 0..2 (box b)   3..5 (box pair)   6..11 (box b, finger f)   12 (fingers)
//     0     static 0  [BB]        b1-b2 [BB]         box - tip   [BB]          tip - tip   [BB]
//     1     static 1  [BB]                           box - hull  [MPR]         hull - hull [MPR]       (groups 13 .. 15: box - hand [wave MPR])
//     2     static 2  [BB]                                                     hull0 - tip1 [MPR]
//     3     static 3  [BB]                                                     tip0 - hull1 [MPR]
// Everything stays in registers and LDS: a pair test emits its contacts into the lane's staging slot of the workgroup's shared area
// (free during this phase), the per-round counts go through the same area, and every lane then moves its contacts to their final
// slots of the environment's compact record list - order: round, then group; deterministic, independent of which environments share
// the workgroup.  The box-box and the MPR jobs of a round run in two separate (non-inlined) functions that derive a lane's job from
// (lane, round) and fetch the shapes from the LDS tables, so that neither carries the other's registers (the combined function spilled
// ~130 doubles per lane and round to scratch: most of the kernel's HBM traffic).
// Results per environment: t[SE_NCON] = contacts kept (<= SK_MAXCON), t[SE_NEED] = 256 when contacts were dropped.
constexpr int SKP_GROUPS = 16;
static_assert(SKP_GROUPS * SK_LANES <= WAVE, "the lane-per-pair collision needs 16 lanes per environment");
struct SkJob { int kind, ba, bb, set, hullA, hullB; double margin; };      // kind: 0 none, 1 box-box, 2 MPR (eight-lane group), 3 MPR against the hand hull (whole wave)
// the job of lane L in a round (L may be another lane: the MPR groups rebuild their owner's job); shapes from the tables of L's environment
__device__ __forceinline__ SkJob sk_job(const StackConsts& kc_, sk_lds_double* smem, const int L, const int round, const unsigned live_mask,
                                        double* RA, double* pA, double* hA, double* RB, double* pB, double* hB) {
  D3IL_STACK_CONSTS(kc_, kc);
  const int e = L % SK_LANES, grp = L / SK_LANES;
  const bool act = grp < SKP_GROUPS && ((live_mask >> e) & 1u) != 0;
  const sk_lds_double* t = sk_env_view(smem, e);
  SkJob j; j.kind = 0; j.ba = 0; j.bb = 0; j.set = 0; j.hullA = 0; j.hullB = 0; j.margin = 0;
  double rsum = 0;
  auto ld = [&](int off, int cnt, double* out) { for (int k = 0; k < cnt; k++) out[k] = t[off + k]; };
  auto box_shape = [&](int b, double* R, double* p, double* h) { ld(ST_BR + 9 * b, 9, R); ld(ST_BP + 3 * b, 3, p); for (int k = 0; k < 3; k++) h[k] = kc.box_half[b][k]; };
  auto tip_shape = [&](int f, double* R, double* p, double* h) { ld(ST_TIPR + 9 * f, 9, R); ld(ST_TIPP + 3 * f, 3, p); for (int k = 0; k < 3; k++) h[k] = kc.tip_half[k]; };
  auto hull_shape = [&](int f, double* R, double* p, double* h) { ld(ST_HULR + 9 * f, 9, R); ld(ST_HULP + 3 * f, 3, p); h[0] = h[1] = h[2] = 0; };
  if (!act) return j;
  if (kc.variant == SKV_ALIGNING) {
    // lane groups 0 .. 4: geom g of the compound body against the static slabs (one per round); 5 .. 9: the rod cylinder against geom g (round 0).
    // A geom's pose: the body's rotation, centre = centre of mass (ST_BP of block 0 in this variant) + R (geom centre - c)
    const int g = grp < AL_NG ? grp : grp - AL_NG;
    if (grp >= 2 * AL_NG || g >= kc.al_ng) return j;
    ld(ST_BR, 9, RB);
    {
      double pc[3]; ld(ST_BP, 3, pc);
      for (int k = 0; k < 3; k++) { pB[k] = pc[k] + RB[3 * k] * kc.al_gpos[g][0] + RB[3 * k + 1] * kc.al_gpos[g][1] + RB[3 * k + 2] * kc.al_gpos[g][2]; hB[k] = kc.al_ghalf[g][k]; }
    }
    if (grp < AL_NG) {
      if (round < kc.ns) {
        const int sidx = round;
        double d[3] = {pB[0] - kc.st_c[sidx][0], pB[1] - kc.st_c[sidx][1], pB[2] - kc.st_c[sidx][2]}, ex = 0;
        for (int i = 0; i < 3; i++) { double loc = kc.st_R[sidx][i] * d[0] + kc.st_R[sidx][3 + i] * d[1] + kc.st_R[sidx][6 + i] * d[2]; double o = fabs(loc) - kc.st_h[sidx][i]; if (o > 0) ex += o * o; }
        const int set = kc.al_set_static[g] + sidx;
        const double rc = kc.al_gr[g] + kc.set[set].margin;
        if (ex <= rc * rc) {
          j.kind = 1; j.ba = SKB_STATIC; j.bb = 0; j.set = set;
          for (int k = 0; k < 9; k++) RA[k] = kc.st_R[sidx][k];
          for (int k = 0; k < 3; k++) { pA[k] = kc.st_c[sidx][k]; hA[k] = kc.st_h[sidx][k]; }
        }
      }
    } else if (round == 0) {      // rod <-> geom: pA = rod centre, RA[0..2] = rod axis, hA = (radius, half length, -); the box geom is geom 1 of the pair
      ld(ST_TIPR + SV_ROD, 3, pA); ld(ST_TIPR + SV_ROD + 3, 3, RA);
      hA[0] = kc.rod_r; hA[1] = kc.rod_h; hA[2] = 0;
      const int set = kc.al_set_rod[g];
      const double w[3] = {pB[0] - pA[0], pB[1] - pA[1], pB[2] - pA[2]};
      const double al = fmin(fmax(w[0] * RA[0] + w[1] * RA[1] + w[2] * RA[2], -kc.rod_h), kc.rod_h);
      const double dd[3] = {w[0] - al * RA[0], w[1] - al * RA[1], w[2] - al * RA[2]}, rc = kc.al_gr[g] + kc.rod_r + kc.set[set].margin;
      if (dot3(dd, dd) <= rc * rc) { j.kind = 4; j.ba = 0; j.bb = SKB_ROD; j.set = set; }
    }
    if (j.kind != 0) j.margin = kc.set[j.set].margin;
    return j;
  }
  if (grp < 3 || (grp >= 13 && round == 0)) {
    // box <-> static slabs: slab 0 in round 0 on the groups 0 .. 2, slab 1 in the SAME round on the groups 13 .. 15 (idle there: their hand job belongs to
    // round 1), further slabs one per later round - a box on the table top lies inside the bounds of both slabs of the table, and every round with a
    // box-box job costs the whole wave one pass through box_box_emit
    const int b = grp < 3 ? grp : grp - 13, sidx = grp < 3 ? (round == 0 ? 0 : round + 1) : 1;
    if (sidx < kc.ns) {
      box_shape(b, RB, pB, hB);
      double d[3] = {pB[0] - kc.st_c[sidx][0], pB[1] - kc.st_c[sidx][1], pB[2] - kc.st_c[sidx][2]}, ex = 0;
      for (int i = 0; i < 3; i++) { double loc = kc.st_R[sidx][i] * d[0] + kc.st_R[sidx][3 + i] * d[1] + kc.st_R[sidx][6 + i] * d[2]; double o = fabs(loc) - kc.st_h[sidx][i]; if (o > 0) ex += o * o; }
      const double rc = kc.box_r[b] + kc.set[SKS_STATIC + sidx].margin;
      if (ex <= rc * rc) {
        j.kind = 1; j.ba = SKB_STATIC; j.bb = b; j.set = SKS_STATIC + sidx;
        for (int k = 0; k < 9; k++) RA[k] = kc.st_R[sidx][k];
        for (int k = 0; k < 3; k++) { pA[k] = kc.st_c[sidx][k]; hA[k] = kc.st_h[sidx][k]; }
      }
    }
  } else if (grp < 6) {
    if (round == 0) {
      const int b1 = grp == 5 ? 1 : 0, b2 = grp == 3 ? 1 : 2;
      box_shape(b1, RA, pA, hA); box_shape(b2, RB, pB, hB);
      j.kind = 1; j.ba = b1; j.bb = b2; j.set = SKS_BOXBOX; rsum = kc.box_r[b1] + kc.box_r[b2];
    }
  } else if (grp < 12) {
    const int b = (grp - 6) >> 1, f = (grp - 6) & 1;
    if (round == 0) { box_shape(b, RA, pA, hA); tip_shape(f, RB, pB, hB); j.kind = 1; j.ba = b; j.bb = SKB_TIP + f; j.set = SKS_BOXTIP; rsum = kc.box_r[b] + kc.tip_r; }
    else if (round == 1) { box_shape(b, RA, pA, hA); hull_shape(f, RB, pB, hB); j.hullB = 1; j.kind = 2; j.ba = b; j.bb = SKB_FINGER + f; j.set = SKS_BOXHULL; rsum = kc.box_r[b] + kc.hull_r; }
  } else if (grp >= 13) {     // box <-> hand: cull with the hull's bounding box; the shapes are rebuilt by the wave-wide MPR stage
    if (round == 1) {
      const int b = grp - 13;
      double Rh[9], ph[3];
      box_shape(b, RA, pA, hA);
      sk_hand_pose(kc, t, t[SE_Q + NARM], Rh, ph);
      if (sk_hand_near(kc, Rh, ph, RA, pA, hA, kc.set[SKS_BOXHAND].margin)) { j.kind = 3; j.ba = b; j.bb = SKB_HAND; j.set = SKS_BOXHAND; }
    }
  } else if (t[ST_AUX] < 0.004) {     // finger <-> finger: only a (nearly) closed gripper (the gaps are q1 + q2 - 1 mm or less)
    if (round == 0) { tip_shape(0, RA, pA, hA); tip_shape(1, RB, pB, hB); j.kind = 1; j.ba = SKB_TIP; j.bb = SKB_TIP + 1; j.set = SKS_TIPTIP; }
    else if (round == 1) { hull_shape(0, RA, pA, hA); hull_shape(1, RB, pB, hB); j.hullA = j.hullB = 1; j.kind = 2; j.ba = SKB_FINGER; j.bb = SKB_FINGER + 1; j.set = SKS_HULLHULL; }
    else if (round == 2) { hull_shape(0, RA, pA, hA); tip_shape(1, RB, pB, hB); j.hullA = 1; j.kind = 2; j.ba = SKB_FINGER; j.bb = SKB_TIP + 1; j.set = SKS_HULLTIP; }
    else { tip_shape(0, RA, pA, hA); hull_shape(1, RB, pB, hB); j.hullB = 1; j.kind = 2; j.ba = SKB_TIP; j.bb = SKB_FINGER + 1; j.set = SKS_HULLTIP; }
  }
  if (j.kind != 0) {
    j.margin = kc.set[j.set].margin;
    if (rsum > 0) {      // bounding spheres about the geom centres (the hull's centre is its mesh centre)
      double cA[3] = {pA[0], pA[1], pA[2]}, cB[3] = {pB[0], pB[1], pB[2]};
      if (j.hullA) for (int k = 0; k < 3; k++) cA[k] += RA[3 * k] * kc.hull_center[0] + RA[3 * k + 1] * kc.hull_center[1] + RA[3 * k + 2] * kc.hull_center[2];
      if (j.hullB) for (int k = 0; k < 3; k++) cB[k] += RB[3 * k] * kc.hull_center[0] + RB[3 * k + 1] * kc.hull_center[1] + RB[3 * k + 2] * kc.hull_center[2];
      const double d[3] = {cB[0] - cA[0], cB[1] - cA[1], cB[2] - cA[2]}, rc = rsum + j.margin;
      if (dot3(d, d) > rc * rc) j.kind = 0;
    }
  }
  return j;
}
// box-box jobs of a round: contacts -> the lane's staging slot (normal[3] | m x (dist, pos[3])); returns m | meta << 8 (0 when the lane has no such job)
__device__ __forceinline__ int sk_round_boxbox(const StackConsts& kc_, sk_lds_double* smem, const int lane, const int round, const unsigned live_mask) {
  double RA[9], pA[3], hA[3], RB[9], pB[3], hB[3];
  const SkJob j = sk_job(kc_, smem, lane, round, live_mask, RA, pA, hA, RB, pB, hB);
  sk_lds_double* stage = smem + lane * SKC_STAGE;
  if (j.kind == 4) {      // rod cylinder against a cube: cyl_box returns {dist, pos, normal from the cube to the rod}
    double r7[7];
    if (!cyl_box(pA, RA, hA[0], hA[1], pB, RB, hB, j.margin, r7)) return 0;
    stage[0] = r7[4]; stage[1] = r7[5]; stage[2] = r7[6];
    stage[3] = r7[0]; stage[4] = r7[1]; stage[5] = r7[2]; stage[6] = r7[3];
    return 1 | ((j.ba | (j.bb << 4) | (j.set << 8)) << 8);
  }
  if (j.kind != 1) return 0;
  int m = 0;
  box_box_emit(pA, RA, hA, pB, RB, hB, j.margin, 8, [&](double dist, const double* pos, const double* nrm) {
    stage[0] = nrm[0]; stage[1] = nrm[1]; stage[2] = nrm[2];
    sk_lds_double* q = stage + 3 + 4 * m;
    q[0] = dist; q[1] = pos[0]; q[2] = pos[1]; q[3] = pos[2];
    m++;
  });
  return m | ((j.ba | (j.bb << 4) | (j.set << 8)) << 8);
}
// support point of the hand hull by the whole wave: lane l evaluates the vertices l, l + 64, ...; maximum and lowest index within 1e-10 of it by
// wave reductions (the vertex sk_support1 finds with its two passes).  R, p: pose of the hand geom (the same values in all lanes).
constexpr int SKH_NV = (SK_MAXHANDV + WAVE - 1) / WAVE;
__device__ __forceinline__ void sk_support_hand_wave(const StackConsts& kc_, const double* R, const double* p, const double* dir, double margin, double* out, const int lane) {
  D3IL_STACK_CONSTS(kc_, kc);
  const double dl[3] = {R[0] * dir[0] + R[3] * dir[1] + R[6] * dir[2], R[1] * dir[0] + R[4] * dir[1] + R[7] * dir[2], R[2] * dir[0] + R[5] * dir[1] + R[8] * dir[2]};
  const int nv = kc.hand_nv;
  double d[SKH_NV], bd = -1e300;
#pragma unroll
  for (int m = 0; m < SKH_NV; m++) {
    const int v = lane + WAVE * m;
    d[m] = v < nv ? kc.hand_v[v][0] * dl[0] + kc.hand_v[v][1] * dl[1] + kc.hand_v[v][2] * dl[2] : -1e300;
    bd = fmax(bd, d[m]);
  }
  bd = sk_wave_max(bd);
  const double thr = bd - 1e-10;
  int best = 1 << 20;
#pragma unroll
  for (int m = SKH_NV - 1; m >= 0; m--) if (d[m] >= thr) best = lane + WAVE * m;
  best = (int)(-sk_wave_max(-(double)best));      // lowest index (exact in a double)
  const double loc[3] = {kc.hand_v[best][0], kc.hand_v[best][1], kc.hand_v[best][2]};
#pragma unroll
  for (int k = 0; k < 3; k++) out[k] = R[3 * k] * loc[0] + R[3 * k + 1] * loc[1] + R[3 * k + 2] * loc[2] + p[k] + 0.5 * margin * dir[k];
}
struct SkShapeR { double R[9], p[3]; int hull; };      // a shape with its pose in registers (the hand geom)
// MPR jobs of a round: up to eight at a time, each on a GROUP of eight lanes (job k of the batch on lanes 8 k .. 8 k + 7).  The group
// rebuilds its owner lane's job (shapes from the LDS tables), the portal iteration runs redundantly on the eight lanes (uniform inside a
// group) and the hull support function is spread over them (sk_support1_group: nine vertices per lane instead of 68 on one lane).
#if defined(D3IL_SK_MPR_NOINLINE)      // experiment of DESIGN section 18.2: the MPR phase as a function of its own (own SGPR / VGPR allocation)
__device__ __attribute__((noinline)) int sk_round_mpr(const StackConsts& kc_, sk_lds_double* smem, const int lane, const int round, const unsigned live_mask) {
#else
__device__ __forceinline__ int sk_round_mpr(const StackConsts& kc_, sk_lds_double* smem, const int lane, const int round, const unsigned live_mask) {
#endif
  D3IL_STACK_CONSTS(kc_, kc);
  int mine_kind = 0, mine_meta = 0;
  {
    double RA[9], pA[3], hA[3], RB[9], pB[3], hB[3];
    const SkJob j = sk_job(kc, smem, lane, round, live_mask, RA, pA, hA, RB, pB, hB);
    mine_kind = j.kind; mine_meta = j.ba | (j.bb << 4) | (j.set << 8);
  }
  int m = 0;
#if defined(D3IL_DEVICE_STATS)
  unsigned long long mp_t0 = wall_clock64();
#endif
  for (unsigned long long pend = __ballot(mine_kind == 2); pend != 0;) {
    const int grp8 = lane >> 3, sub = lane & 7;
#if defined(D3IL_DEVICE_STATS)
    if (lane == 0 && blockIdx.x == 0) atomicAdd(&g_dev_stats[8], 1ull);
#endif
    int owner = -1;      // owner lane of this group's job (-1: no job in this batch)
    unsigned long long batch = 0, rest = pend;
    for (int k = 0; k < WAVE / SKG && rest != 0; k++) { const int L = __builtin_ctzll(rest); rest &= rest - 1; batch |= 1ull << L; if (k == grp8) owner = L; }
    pend = rest;
    double r7[7] = {0, 0, 0, 0, 0, 0, 0};
    int hit = 0;
    if (owner >= 0) {
      // the owner's job, shapes as LDS table offsets (groups 6 .. 11: box b - hull f; group 12: hull - hull, hull0 - tip1, tip0 - hull1)
      const int oe = owner % SK_LANES, og = owner / SK_LANES;
      const sk_lds_double* to = sk_env_view(smem, oe);
      SkShapeL A, B;
      int set;
      auto box_l = [&](int bx, SkShapeL& S) { S.R = to + ST_BR + 9 * bx; S.p = to + ST_BP + 3 * bx; for (int k = 0; k < 3; k++) S.half[k] = kc.box_half[bx][k]; S.hull = 0; };
      auto tip_l = [&](int f, SkShapeL& S) { S.R = to + ST_TIPR + 9 * f; S.p = to + ST_TIPP + 3 * f; for (int k = 0; k < 3; k++) S.half[k] = kc.tip_half[k]; S.hull = 0; };
      auto hull_l = [&](int f, SkShapeL& S) { S.R = to + ST_HULR + 9 * f; S.p = to + ST_HULP + 3 * f; S.half[0] = S.half[1] = S.half[2] = 0; S.hull = 1; };
      if (og < 12) { box_l((og - 6) >> 1, A); hull_l((og - 6) & 1, B); set = SKS_BOXHULL; }
      else if (round == 1) { hull_l(0, A); hull_l(1, B); set = SKS_HULLHULL; }
      else if (round == 2) { hull_l(0, A); tip_l(1, B); set = SKS_HULLTIP; }
      else { tip_l(0, A); hull_l(1, B); set = SKS_HULLTIP; }
      const double um = kc.set[set].margin;
      double hv[SKG_NV][3];      // this lane's share of the finger-hull vertices (every job of these groups involves the finger hull)
#pragma unroll
      for (int m2 = 0; m2 < SKG_NV; m2++) { const int v = sub + SKG * m2; const int vv = v < kc.hull_nv ? v : 0; hv[m2][0] = kc.hull_v[vv][0]; hv[m2][1] = kc.hull_v[vv][1]; hv[m2][2] = kc.hull_v[vv][2]; }
      hit = sk_mpr_t(kc, A, B, um, r7, [&](const double* dir, SkPt& pt) {
        const double nd[3] = {-dir[0], -dir[1], -dir[2]};
#if defined(D3IL_SK_NO_PRELOAD)
        sk_support1_group_l(kc, A, dir, um, pt.v1, sub); sk_support1_group_l(kc, B, nd, um, pt.v2, sub);
#else
        sk_support1_group_lp(kc, A, hv, dir, um, pt.v1, sub); sk_support1_group_lp(kc, B, hv, nd, um, pt.v2, sub);
#endif
#pragma unroll
        for (int k = 0; k < 3; k++) pt.v[k] = pt.v1[k] - pt.v2[k];
      }) ? 1 : 0;
    }
    // results back to the owner lanes: owner L reads lane 8 * (rank of L in the batch)
    const bool mine = ((batch >> lane) & 1ull) != 0;
    const int from = mine ? SKG * __popcll(batch & ((1ull << lane) - 1ull)) : lane;
    SK_CONVERGE();
    const int ghit = __shfl(hit, from);
    double g7[7];
#pragma unroll
    for (int k = 0; k < 7; k++) g7[k] = __shfl(r7[k], from);
    SK_CONVERGE();
    if (mine && ghit) {
      sk_lds_double* stage = smem + lane * SKC_STAGE;
      stage[0] = g7[4]; stage[1] = g7[5]; stage[2] = g7[6];
      stage[3] = g7[0]; stage[4] = g7[1]; stage[5] = g7[2]; stage[6] = g7[3];
      m = 1;
    }
  }
  // box <-> hand jobs (rare: only inside the hull's bounding box): one at a time by the whole wave, the 773 hull vertices spread over the lanes
#if defined(D3IL_DEVICE_STATS)
  if (lane == 0 && blockIdx.x == 0) { const unsigned long long t_ = wall_clock64(); atomicAdd(&g_dev_stats[11], t_ - mp_t0); mp_t0 = t_; }
#endif
  for (unsigned long long pendh = __ballot(mine_kind == 3); pendh != 0; pendh &= pendh - 1) {
    const int owner = __builtin_ctzll(pendh);
#if defined(D3IL_DEVICE_STATS)
    if (lane == 0 && blockIdx.x == 0) atomicAdd(&g_dev_stats[9], 1ull);
#endif
    const int oe = owner % SK_LANES, bx = owner / SK_LANES - 13;
    const sk_lds_double* to = sk_env_view(smem, oe);
    SkShapeL A;
    A.R = to + ST_BR + 9 * bx; A.p = to + ST_BP + 3 * bx; for (int k = 0; k < 3; k++) A.half[k] = kc.box_half[bx][k]; A.hull = 0;
    SkShapeR B;
    sk_hand_pose(kc, to, to[SE_Q + NARM], B.R, B.p); B.hull = 2;
    const double um = kc.set[SKS_BOXHAND].margin;
    double r7[7] = {0, 0, 0, 0, 0, 0, 0};
    const bool hit = sk_mpr_t(kc, A, B, um, r7, [&](const double* dir, SkPt& pt) {
      const double nd[3] = {-dir[0], -dir[1], -dir[2]};
      sk_support1_group_l(kc, A, dir, um, pt.v1, lane & 7);      // a box: no vertex table, every lane evaluates it
      sk_support_hand_wave(kc, B.R, B.p, nd, um, pt.v2, lane);
#if defined(D3IL_DEVICE_STATS)
      if (lane == 0 && blockIdx.x == 0) atomicAdd(&g_dev_stats[10], 1ull);
#endif
#pragma unroll
      for (int k = 0; k < 3; k++) pt.v[k] = pt.v1[k] - pt.v2[k];
    });
    if (lane == owner && hit) {
      sk_lds_double* stage = smem + lane * SKC_STAGE;
      stage[0] = r7[4]; stage[1] = r7[5]; stage[2] = r7[6];
      stage[3] = r7[0]; stage[4] = r7[1]; stage[5] = r7[2]; stage[6] = r7[3];
      m = 1;
    }
  }
#if defined(D3IL_DEVICE_STATS)
  if (lane == 0 && blockIdx.x == 0) { const unsigned long long t_ = wall_clock64(); atomicAdd(&g_dev_stats[12], t_ - mp_t0); }
#endif
  return m | (mine_meta << 8);
}
__device__ __forceinline__ void sk_collide_coop(const StackConsts& kc_, sk_lds_double* smem, const int lane, const unsigned live_mask, double* dbg = nullptr) {
  D3IL_STACK_CONSTS(kc_, kc);
  const int e = lane % SK_LANES, grp = lane / SK_LANES;
  const bool act = grp < SKP_GROUPS && ((live_mask >> e) & 1u) != 0;
  sk_lds_double* t = sk_env_view(smem, e);
  sk_lds_double* stage = smem + (grp < SKP_GROUPS ? lane : 0) * SKC_STAGE;
  int tot = 0;      // contacts of this environment so far (the same value in all its lanes)
#if defined(D3IL_DEVICE_STATS)
  unsigned long long skp_t0 = wall_clock64();      // lane 0 (environment 0, group 0) times the phases of the whole wave: slots 14 box-box, 15 MPR, 13 bookkeeping
#define SKP_TOC(slot) do { unsigned long long t_ = wall_clock64(); if (lane == 0 && blockIdx.x == 0) atomicAdd(&g_dev_stats[16 + (slot)], t_ - skp_t0); skp_t0 = t_; } while (0)
#else
#define SKP_TOC(slot) ((void)0)
#endif
  // rounds: 0 = slabs 0 and 1, box pairs, box <-> tip, tip <-> tip; 1 = slab 2, box <-> hull / hand, hull <-> hull; 2, 3 = further slabs and the remaining
  // finger <-> finger pairs - those two rounds are run only when a slab or a (nearly) closed gripper of the workgroup needs them (wave-uniform)
  int n_rounds = kc.ns;
  if (kc.variant == SKV_STACKING) {
    const bool closed = __any(act && t[ST_AUX] < 0.004);
    n_rounds = closed ? 4 : 2;
    if (kc.ns - 1 > n_rounds) n_rounds = kc.ns - 1;
  }
  for (int round = 0; round < n_rounds; round++) {
    int r = sk_round_boxbox(kc, smem, lane, round, live_mask);
    SKP_TOC(14);
    if (round >= 1) { const int r2 = sk_round_mpr(kc, smem, lane, round, live_mask); if ((r2 & 255) != 0) r = r2; }
    SKP_TOC(15);
    const int m = r & 255;
#if defined(D3IL_DEVICE_STATS)
    if (dbg && grp < SKP_GROUPS) dbg[(size_t)e * SG_SIZE + 300 + round * 16 + grp] = (double)r + 0.5 * (t[ST_AUX] < 0.004 ? 1 : 0);
#endif
    // counts through the staging slots, then every lane moves its contacts to their final records
    if (grp < SKP_GROUPS) stage[SKC_STAGE - 1] = (double)m;
    __syncthreads();
    int off = tot, all = 0;
    for (int g = 0; g < SKP_GROUPS; g++) { const int c = (int)smem[(g * SK_LANES + e) * SKC_STAGE + SKC_STAGE - 1]; if (g < grp) off += c; all += c; }
    if (act && m > 0) {
      const double meta = (double)(r >> 8);
      const double n0 = stage[0], n1 = stage[1], n2 = stage[2];
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const int slot = off + i;
        if (i >= m || slot >= SK_MAXCON) continue;
        const sk_lds_double* q = stage + 3 + 4 * i;
        sk_lds_double* rr = t + SE_REC + slot * SREC2;
        rr[0] = q[1]; rr[1] = q[2]; rr[2] = q[3]; rr[3] = n0; rr[4] = n1; rr[5] = n2; rr[6] = q[0]; rr[7] = meta;
      }
    }
    tot += all;
    __syncthreads();
    SKP_TOC(13);
  }
  if (lane < SK_LANES && act) { t[SE_NCON] = (double)(tot < SK_MAXCON ? tot : SK_MAXCON); t[SE_NEED] = tot > SK_MAXCON ? 256.0 : 0.0; }
  __syncthreads();
}
#undef SKP_TOC
#endif

// mj_Euler: implicit in the finger-joint damping, (M + h B) qacc = M x on the arm block; the arm mass matrix is still in the t area
template <bool WARM_LDS, class C>
D3IL_HD void stack_substep_post(const C& c0, const StackConsts& kc_, StackState& ss, const StackScratch sc) {
  D3IL_REFRESH(c0, c);
  D3IL_STACK_CONSTS(kc_, kc);
  EnvState& st = ss.arm;
  const double h = c.timestep;
  SK_TIC;
  if (!WARM_LDS) for (int i = 0; i < SK_NV; i++) ss.warm[i] = SL(ST_X + i);
  st.flags |= SKF_WARM_VALID;
  {
    double Mh[45], rhs[NDOF], L[45], d[NDOF], id[NDOF];
    for (int k = 0; k < 45; k++) Mh[k] = SL(ST_M + k);
    for (int a = 0; a < NDOF; a++) { double s = 0; for (int k = 0; k < NDOF; k++) s += Mh[a >= k ? tri(a, k) : tri(k, a)] * SL(ST_X + SK_ARM0 + k); rhs[a] = s; }
    for (int k = 0; k < NFING; k++) Mh[tri(NARM + k, NARM + k)] += h * c.f_damping[k];
    if (!ldl9(Mh, L, d, id)) st.flags |= F_SOLVER_FAIL;
    ldl9_solve(L, id, rhs);
    for (int k = 0; k < NDOF; k++) { st.v[k] += h * rhs[k]; st.q[k] += h * st.v[k]; }
  }
  for (int b = 0; b < SK_NB; b++) {
    double acc[6];
    for (int k = 0; k < 6; k++) acc[k] = SL(ST_X + 6 * b + k);
    if (kc.variant == SKV_ALIGNING && b == 0) {
      // the state read from the tables and the solution are in centre-of-mass coordinates (stack_pre_kin): back to MuJoCo's body origin
      // before mj_Euler: p_o = p_c - R c, v_o = v_c - R (w x c), a_o = a_c - R (alpha x c) - w_w x (w_w x R c)
      double R[9], t3[3], wc[3], ac[3];
      for (int k = 0; k < 9; k++) R[k] = SL(ST_BR + k);
      mulE(R, kc.al_c, t3); for (int k = 0; k < 3; k++) ss.box[0].pos[k] -= t3[k];
      cross3(ss.box[0].vel + 3, kc.al_c, wc); mulE(R, wc, t3); for (int k = 0; k < 3; k++) ss.box[0].vel[k] -= t3[k];
      cross3(acc + 3, kc.al_c, ac); mulE(R, ac, t3); for (int k = 0; k < 3; k++) acc[k] -= t3[k] + SL(ST_TIPR + SV_CEN + k);
    }
    cube_integrate(ss.box[b], acc, h);
  }
  SK_TOC(5);
  (void)kc;
}
// the sub-step on one lane (host build, reset kernel)
template <class C>
D3IL_NOINLINE inline void stack_physics_substep(const C& c0, const StackConsts& kc_, StackState& ss, const StackScratch sc, const double* tau, const double* ffing) {
  D3IL_STACK_CONSTS(kc_, kc);
  int ncon; bool any_lim;
  stack_substep_pre<false>(c0, kc, ss, sc, tau, ffing, ncon, any_lim);
  SK_TIC;
  // reference accelerations and regularisation of the contact rows (mj_makeImpedance, elliptic cones)
  for (int ci = 0; ci < ncon; ci++) sk_contact_dispatch<1>(kc, sc, ci, 1);
  if (ncon > 0 || any_lim) {
    const bool ok = sk_solve(kc, sc, ncon, any_lim, (ss.arm.flags & SKF_WARM_VALID) != 0); SG(SG_DIAG + 2) = ok ? 1.0 : 0.0; if (!ok) ss.arm.flags |= F_SOLVER_FAIL;
  }
  SK_TOC(4);
  stack_substep_post<false>(c0, kc, ss, sc);
}

// ------------------------------------------------------------------------------------------------ env level (stacking.py)
D3IL_HD void stack_obs(const StackState& ss, float* obs) {     // stacking.py:228-277: (x, y, z, tan yaw) x 3
  for (int b = 0; b < SK_NB; b++) {
    obs[4 * b] = (float)ss.box[b].pos[0]; obs[4 * b + 1] = (float)ss.box[b].pos[1]; obs[4 * b + 2] = (float)ss.box[b].pos[2];
    obs[4 * b + 3] = (float)push_tan_yaw(ss.box[b].quat);
  }
}
D3IL_HD bool stack_success(const StackConsts& kc_, const StackState& ss) {    // _check_early_termination, stacking.py:425-447
  D3IL_STACK_CONSTS(kc_, kc);
  double dz = fmin(fabs(ss.box[0].pos[2] - ss.box[1].pos[2]), fmin(fabs(ss.box[0].pos[2] - ss.box[2].pos[2]), fabs(ss.box[1].pos[2] - ss.box[2].pos[2])));
  bool ok = dz > 0.03;
  for (int b = 0; b < SK_NB; b++) {
    double dx = ss.box[b].pos[0] - kc.target[0], dy = ss.box[b].pos[1] - kc.target[1];
    ok = ok && sqrt(dx * dx + dy * dy) <= kc.min_dist;
  }
  return ok;
}
D3IL_HD void stack_check_mode(const StackConsts& kc_, StackState& ss, double* mean_dist) {   // check_mode, stacking.py:395-419
  D3IL_STACK_CONSTS(kc_, kc);
  unsigned fl = ss.arm.flags;
  int n = (int)(fl & SKF_NMODE_MASK);
  double d[3], md = 0;
  for (int b = 0; b < SK_NB; b++) { double dx = ss.box[b].pos[0] - kc.target[0], dy = ss.box[b].pos[1] - kc.target[1]; d[b] = sqrt(dx * dx + dy * dy); md += d[b]; }
  *mean_dist = md / 3;
  for (int i = 0; i < n; i++) d[(fl >> (SKF_IND_SHIFT + 2 * i)) & 3u] = 100000;
  int mi = 0;
  for (int b = 1; b < SK_NB; b++) if (d[b] < d[mi]) mi = b;
  if (d[mi] <= kc.min_dist && n < 3) {
    fl = (fl & ~SKF_NMODE_MASK) | (unsigned)(n + 1);
    fl = (fl & ~(3u << (SKF_IND_SHIFT + 2 * n))) | ((unsigned)mi << (SKF_IND_SHIFT + 2 * n));
  }
  ss.arm.flags = fl;
}
D3IL_HD int stack_mode_code(unsigned flags) { return (int)(flags & 0xFFu); }   // n | letters << 2 (0 r, 1 g, 2 b): info['mode'] as an integer

// joint PD with qd_des = 0 (module comment of the oracle's Stacking section) + finger control
template <class C>
D3IL_HD void stack_control(const C& c, const EnvState& st, const double* q_des, double set_width, bool grasp, double* tau, double* ff) {
  double zero[NARM] = {0, 0, 0, 0, 0, 0, 0};
  push_control(c, st, q_des, zero, set_width, grasp, tau, ff);
}
// env.step in three parts: stack_env_begin (gripper command, observation and done flag BEFORE the physics), n_substeps x (control +
// sub-step), stack_env_end (counter, success, mode)
D3IL_HD bool stack_env_begin(const StackConsts& kc_, StackState& ss, const double* action, float* obs, unsigned char* done, int max_steps) {
  D3IL_STACK_CONSTS(kc_, kc);
  const bool open = action[7] > kc.grip_thresh;
  stack_obs(ss, obs);
  bool fin = (ss.arm.flags & F_TERMINATED) != 0;
  if (!fin && stack_success(kc, ss)) { ss.arm.flags |= F_TERMINATED; fin = true; }
  if (!fin && ss.arm.step >= max_steps - 1) fin = true;
  *done = fin ? 1 : 0;
  return open;
}
D3IL_HD void stack_env_end(const StackConsts& kc_, StackState& ss, double* mean_dist) {
  D3IL_STACK_CONSTS(kc_, kc);
  ss.arm.step += 1;
  if (stack_success(kc, ss)) ss.arm.flags |= F_SUCCESS | F_TERMINATED; else ss.arm.flags &= ~F_SUCCESS;
  stack_check_mode(kc, ss, mean_dist);
}
template <class C>
D3IL_HD void stack_env_step(const C& c, const StackConsts& kc_, StackState& ss, const StackScratch sc, const double* action, float* obs, unsigned char* done,
                            double* mean_dist, int n_substeps, int max_steps) {
  D3IL_STACK_CONSTS(kc_, kc);
  const bool open = stack_env_begin(kc, ss, action, obs, done, max_steps);
  const double width = open ? 0.04 : 0.0;
  for (int s = 0; s < n_substeps; s++) {
    double tau[NARM], ff[NFING];
    stack_control(c, ss.arm, action, width, !open, tau, ff);
    stack_physics_substep(c, kc, ss, sc, tau, ff);
  }
  stack_env_end(kc, ss, mean_dist);
}
// reset(random=False, context): scene.reset + beam to init_qpos + open_fingers + contexts + one sub-step (stacking.py:449-481)
template <class C>
D3IL_HD void stack_env_reset(const C& c, const StackConsts& kc_, StackState& ss, const StackScratch sc, const double* init_qpos, const double* ctx, float* obs) {
  D3IL_STACK_CONSTS(kc_, kc);
  EnvState& st = ss.arm;
  for (int k = 0; k < NDOF; k++) { st.q[k] = k < NARM ? init_qpos[k] : 0.0; st.v[k] = 0; }
  st.flags = 0; st.step = 0;
  for (int i = 0; i < SK_NV; i++) ss.warm[i] = 0;
  {   // mj_forward at the beamed pose: qfrc_bias and TCP of that pass are what the first controller call reads
    DynOut dyn;
    dynamics(c, st.q, st.v, dyn);
    for (int k = 0; k < NARM; k++) st.bias[k] = dyn.bias[k];
    double t[3]; mulE(dyn.R7, c.tcp7, t);
    for (int k = 0; k < 3; k++) st.tcp[k] = dyn.p7[k] + t[k];
  }
  for (int b = 0; b < SK_NB; b++) {
    for (int k = 0; k < 3; k++) ss.box[b].pos[k] = ctx[7 * b + k];
    for (int k = 0; k < 4; k++) ss.box[b].quat[k] = ctx[7 * b + 3 + k];
    for (int k = 0; k < 6; k++) ss.box[b].vel[k] = 0;
  }
  double tau[NARM], ff[NFING];
  stack_control(c, st, init_qpos, 0.04, false, tau, ff);
  stack_physics_substep(c, kc, ss, sc, tau, ff);
  stack_obs(ss, obs);
  (void)kc;
}

// ------------------------------------------------------------------------------------------------ constants from the blob (host)
D3IL_HOSTFN inline int build_stack_consts(const d3il_model_blob& m, const PandaConsts& pcst, StackConsts& kc, const char** err) {
  using hostmath::Xf; using hostmath::identity; using hostmath::compose; using hostmath::mv; using hostmath::solref_kb; using hostmath::clamp_solimp;
  std::memset(&kc, 0, sizeof kc);
  if (m.n_obj != SK_NB) { *err = "stacking needs three task objects"; return -1; }
  if (m.nmesh < 1) { *err = "stacking needs the finger hull (blob meshes)"; return -1; }
  kc.nb = SK_NB;
  auto obj_geom = [&](int body) { for (int g = 0; g < m.ngeom; g++) if (m.geom_body[g] == body && m.geom_contype[g]) return g; return -1; };
  int gb[SK_NB];
  for (int b = 0; b < SK_NB; b++) {
    int bd = m.obj_body[b]; gb[b] = obj_geom(bd);
    if (gb[b] < 0 || m.geom_type[gb[b]] != D3IL_GEOM_BOX) { *err = "task objects must be boxes"; return -1; }
    for (int k = 0; k < 3; k++) { kc.box_half[b][k] = m.geom_size[gb[b]][k]; kc.box_inertia[b][k] = m.body_inertia[bd][k]; if (m.geom_pos[gb[b]][k] != 0) { *err = "boxes must be centred on their bodies"; return -1; } }
    if (m.body_iquat[bd][0] != 1.0) { *err = "box inertial frames must be the body frames"; return -1; }
    kc.box_mass[b] = m.body_mass[bd]; kc.box_invw[b] = 1.0 / kc.box_mass[b];
    if (b > 0 && gb[b] < gb[b - 1]) { *err = "unexpected geom order"; return -1; }
  }
  // world transforms of all bodies at q = 0
  static thread_local Xf X0[D3IL_MAXBODY];
  X0[0] = identity();
  for (int b = 1; b < m.nbody; b++) { Xf l; quat2mat(m.body_quat[b], l.R); std::memcpy(l.p, m.body_pos[b], sizeof l.p); X0[b] = compose(X0[m.body_parent[b]], l); }
  auto rel = [&](int a, int b) {
    Xf inv; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) inv.R[3 * i + j] = X0[a].R[3 * j + i];
    double t[3]; mv(inv.R, X0[a].p, t); for (int k = 0; k < 3; k++) inv.p[k] = -t[k];
    return compose(inv, X0[b]);
  };
  auto weld_root = [&](int b) { while (b > 0 && m.body_jntnum[b] == 0) b = m.body_parent[b]; return b; };
  auto mix = [&](int g1, int g2, StackSet& s) {   // mj_contactParam: priority, else solmix average / max friction / max condim
    int src = m.geom_priority[g1] > m.geom_priority[g2] ? g1 : (m.geom_priority[g2] > m.geom_priority[g1] ? g2 : -1);
    double sr[2], si[5];
    if (src >= 0) { for (int k = 0; k < 2; k++) sr[k] = m.geom_solref[src][k]; for (int k = 0; k < 5; k++) si[k] = m.geom_solimp[src][k]; for (int k = 0; k < 3; k++) s.fric[k] = m.geom_friction[src][k]; s.dim = m.geom_condim[src]; }
    else {
      double s1 = m.geom_solmix[g1], s2 = m.geom_solmix[g2], w = s1 / (s1 + s2);
      for (int k = 0; k < 2; k++) sr[k] = w * m.geom_solref[g1][k] + (1 - w) * m.geom_solref[g2][k];
      for (int k = 0; k < 5; k++) si[k] = w * m.geom_solimp[g1][k] + (1 - w) * m.geom_solimp[g2][k];
      for (int k = 0; k < 3; k++) s.fric[k] = std::fmax(m.geom_friction[g1][k], m.geom_friction[g2][k]);
      s.dim = m.geom_condim[g1] > m.geom_condim[g2] ? m.geom_condim[g1] : m.geom_condim[g2];
    }
    solref_kb(sr, si, m.timestep, &s.K, &s.B);
    clamp_solimp(si, s.solimp);
    s.margin = std::fmax(m.geom_margin[g1], m.geom_margin[g2]) - std::fmax(m.geom_gap[g1], m.geom_gap[g2]);
    return (s.dim == 3 || s.dim == 4) ? 0 : -1;
  };
  // static boxes under the workspace (as build_gen_consts: the table-top footprint without its rim is the modelled workspace)
  double rmax = 0;
  for (int b = 0; b < SK_NB; b++) rmax = std::fmax(rmax, std::sqrt(kc.box_half[b][0] * kc.box_half[b][0] + kc.box_half[b][1] * kc.box_half[b][1] + kc.box_half[b][2] * kc.box_half[b][2]));
  int ns = 0, table = -1;
  double wlo[3] = {0, 0, 0}, whi[3] = {0, 0, 0};
  for (int pass = 0; pass < 2; pass++)
  for (int g = 0; g < m.ngeom; g++) {
    if (m.geom_type[g] != D3IL_GEOM_BOX) continue;
    if (!((m.geom_contype[g] & m.geom_conaffinity[gb[0]]) || (m.geom_contype[gb[0]] & m.geom_conaffinity[g]))) continue;
    if (weld_root(m.geom_body[g]) != 0) continue;
    Xf gl; quat2mat(m.geom_quat[g], gl.R); std::memcpy(gl.p, m.geom_pos[g], sizeof gl.p);
    Xf xg = compose(X0[m.geom_body[g]], gl);
    double ext[3];
    for (int i = 0; i < 3; i++) ext[i] = std::fabs(xg.R[3 * i]) * m.geom_size[g][0] + std::fabs(xg.R[3 * i + 1]) * m.geom_size[g][1] + std::fabs(xg.R[3 * i + 2]) * m.geom_size[g][2];
    if (pass == 0) {
      if (std::fabs(m.geom_size[g][0] - 0.49) < 1e-12 && std::fabs(m.geom_size[g][1] - 0.98) < 1e-12 && std::fabs(m.geom_size[g][2] - 0.001) < 1e-12) {
        table = g;
        // the aluminium profiles around the table edge reach ~2.5 cm into the footprint: a box centre stays 3 cm + its circumradius inside
        for (int i = 0; i < 2; i++) { wlo[i] = xg.p[i] - ext[i] + 0.03 + rmax; whi[i] = xg.p[i] + ext[i] - 0.03 - rmax; kc.ws_lo[i] = wlo[i]; kc.ws_hi[i] = whi[i]; wlo[i] -= rmax; whi[i] += rmax; }
        wlo[2] = xg.p[2] + ext[2] - rmax;
      }
      continue;
    }
    if (table < 0) { *err = "table slab not found"; return -1; }
    if (xg.p[0] + ext[0] < wlo[0] || xg.p[0] - ext[0] > whi[0] || xg.p[1] + ext[1] < wlo[1] || xg.p[1] - ext[1] > whi[1] || xg.p[2] + ext[2] < wlo[2]) continue;
    if (ns >= SK_MAXNS) { *err = "too many static boxes"; return -1; }
    if (g > gb[0]) { *err = "static geoms must precede the boxes"; return -1; }
    for (int k = 0; k < 3; k++) { kc.st_c[ns][k] = xg.p[k]; kc.st_h[ns][k] = m.geom_size[g][k]; }
    for (int k = 0; k < 9; k++) kc.st_R[ns][k] = xg.R[k];
    if (mix(g, gb[0], kc.set[SKS_STATIC + ns])) { *err = "unsupported contact dimension"; return -1; }
    ns++;
  }
  kc.ns = ns;
  // finger geoms
  int link7 = m.jnt_body[m.act_jnt[NARM - 1]];
  int ghull[NFING] = {-1, -1}, gtip[NFING] = {-1, -1};
  for (int f = 0; f < NFING; f++) {
    int fb = m.jnt_body[m.act_jnt[NARM + f]];
    for (int g = 0; g < m.ngeom; g++) {
      if (weld_root(m.geom_body[g]) != fb || !m.geom_contype[g]) continue;
      if (m.geom_type[g] == D3IL_GEOM_MESH && m.geom_mesh[g] >= 0) ghull[f] = g;
      if (m.geom_type[g] == D3IL_GEOM_BOX) gtip[f] = g;
    }
    if (ghull[f] < 0 || gtip[f] < 0) { *err = "finger hull / tip geoms not found"; return -1; }
    for (int g = 0; g < 2; g++) {
      int gg = g ? ghull[f] : gtip[f];
      Xf gl; quat2mat(m.geom_quat[gg], gl.R); std::memcpy(gl.p, m.geom_pos[gg], sizeof gl.p);
      Xf xg = compose(rel(link7, m.geom_body[gg]), gl);
      std::memcpy(g ? kc.hull_R[f] : kc.tip_R[f], xg.R, sizeof xg.R); std::memcpy(g ? kc.hull_p[f] : kc.tip_p[f], xg.p, sizeof xg.p);
    }
    for (int k = 0; k < 3; k++) if (m.geom_size[gtip[f]][k] != m.geom_size[gtip[0]][k]) { *err = "finger tips must be identical"; return -1; }
  }
  if (!(gb[SK_NB - 1] < ghull[0] && ghull[0] < gtip[0] && gtip[0] < ghull[1] && ghull[1] < gtip[1])) { *err = "unexpected finger geom order"; return -1; }
  for (int k = 0; k < 3; k++) kc.tip_half[k] = m.geom_size[gtip[0]][k];
  int mi = m.geom_mesh[ghull[0]];
  if (m.geom_mesh[ghull[1]] != mi || m.mesh_nvert[mi] > SK_MAXHV) { *err = "finger hulls must share one mesh"; return -1; }
  kc.hull_nv = m.mesh_nvert[mi];
  for (int i = 0; i < kc.hull_nv; i++) for (int k = 0; k < 3; k++) kc.hull_v[i][k] = m.mesh_vert[mi][i][k];
  for (int k = 0; k < 3; k++) kc.hull_center[k] = m.mesh_center[mi][k];
  {   // bounding radii (sqrt of the same sums the per-step code used to form)
    double r2 = 0;
    for (int i = 0; i < kc.hull_nv; i++) { double d[3] = {kc.hull_v[i][0] - kc.hull_center[0], kc.hull_v[i][1] - kc.hull_center[1], kc.hull_v[i][2] - kc.hull_center[2]}; r2 = std::fmax(r2, dot3(d, d)); }
    kc.hull_r = std::sqrt(r2);
    kc.tip_r = std::sqrt(kc.tip_half[0] * kc.tip_half[0] + kc.tip_half[1] * kc.tip_half[1] + kc.tip_half[2] * kc.tip_half[2]);
    for (int b = 0; b < SK_NB; b++) kc.box_r[b] = std::sqrt(kc.box_half[b][0] * kc.box_half[b][0] + kc.box_half[b][1] * kc.box_half[b][1] + kc.box_half[b][2] * kc.box_half[b][2]);
  }
  if (mix(gb[0], gb[1], kc.set[SKS_BOXBOX]) || mix(gb[0], ghull[0], kc.set[SKS_BOXHULL]) || mix(gb[0], gtip[0], kc.set[SKS_BOXTIP]) ||
      mix(ghull[0], ghull[1], kc.set[SKS_HULLHULL]) || mix(ghull[0], gtip[1], kc.set[SKS_HULLTIP]) || mix(gtip[0], gtip[1], kc.set[SKS_TIPTIP])) { *err = "unsupported contact dimension"; return -1; }
  // translational body_invweight0 of the finger and finger-tip bodies at qpos0: mean diagonal of Jp M^-1 Jp' at the body's COM
  {
    double q[NDOF] = {0}, v[NDOF] = {0};
    DynOut dyn;
    dynamics(pcst, q, v, dyn);
    double L[45], d[NDOF], id[NDOF], Minv[NDOF][NDOF];
    ldl9(dyn.M, L, d, id);
    for (int col = 0; col < NDOF; col++) { double e[NDOF] = {0}; e[col] = 1; ldl9_solve(L, id, e); for (int r = 0; r < NDOF; r++) Minv[r][col] = e[r]; }
    double sn[NARM], cs[NARM], R7[9], p7[3], ax[NARM][3], og[NARM][3];
    for (int i = 0; i < NARM; i++) { sn[i] = 0; cs[i] = 1; }
    world_chain(pcst, sn, cs, R7, p7, ax, og);
    for (int f = 0; f < NFING; f++) {
      int fb = m.jnt_body[m.act_jnt[NARM + f]];
      for (int which = 0; which < 2; which++) {
        int body = which ? m.geom_body[gtip[f]] : fb;
        Xf xb = rel(link7, body);
        double cl[3], cw[3];
        mv(xb.R, m.body_ipos[body], cl); for (int k = 0; k < 3; k++) cl[k] += xb.p[k];
        mulE(R7, cl, cw); for (int k = 0; k < 3; k++) cw[k] += p7[k];
        double J[3][NDOF];
        for (int k = 0; k < NARM; k++) { double dd[3] = {cw[0] - og[k][0], cw[1] - og[k][1], cw[2] - og[k][2]}, col[3]; cross3(ax[k], dd, col); for (int r = 0; r < 3; r++) J[r][k] = col[r]; }
        double axw[3]; mulE(R7, pcst.f_axis[f], axw);
        for (int g = 0; g < NFING; g++) for (int r = 0; r < 3; r++) J[r][NARM + g] = g == f ? axw[r] : 0.0;
        double tr = 0;
        for (int r = 0; r < 3; r++) for (int a = 0; a < NDOF; a++) for (int b = 0; b < NDOF; b++) tr += J[r][a] * Minv[a][b] * J[r][b];
        (which ? kc.invw_tip[f] : kc.invw_finger[f]) = std::fmax(1e-15, tr / 3);
      }
    }
  }
  kc.impratio = m.impratio;
  for (int k = 0; k < 3; k++) kc.target[k] = m.task_f[k];
  kc.min_dist = m.task_f[3]; kc.grip_thresh = m.task_f[4];
  {   // hand geom (panda_hand:geom2, mesh handv): frame in the link-7 frame, hull, bounding box of the hull (task_f[5..10]), contact parameters, invweight
    int hand = m.body_parent[m.jnt_body[m.act_jnt[NARM]]];
    int gh = -1;
    for (int g = 0; g < m.ngeom; g++) if (m.geom_body[g] == hand && m.geom_type[g] == D3IL_GEOM_MESH && m.geom_contype[g] && m.geom_mesh[g] >= 0) gh = g;
    if (gh < 0) { *err = "stacking needs the hand hull (blob meshes: handv)"; return -1; }
    Xf gl; quat2mat(m.geom_quat[gh], gl.R); std::memcpy(gl.p, m.geom_pos[gh], sizeof gl.p);
    Xf xh = compose(rel(link7, hand), gl);
    std::memcpy(kc.hand_R, xh.R, sizeof xh.R); std::memcpy(kc.hand_p, xh.p, sizeof xh.p);
    for (int k = 0; k < 3; k++) { kc.hand_lo[k] = m.task_f[5 + k]; kc.hand_hi[k] = m.task_f[8 + k]; kc.f_axis0[k] = pcst.f_axis[0][k]; }
    const int hm = m.geom_mesh[gh];
    if (m.mesh_nvert[hm] > SK_MAXHANDV) { *err = "hand hull too large"; return -1; }
    kc.hand_nv = m.mesh_nvert[hm];
    double r2 = 0;
    for (int i = 0; i < kc.hand_nv; i++) {
      double d[3];
      for (int k = 0; k < 3; k++) { kc.hand_v[i][k] = m.mesh_vert[hm][i][k]; d[k] = kc.hand_v[i][k] - m.mesh_center[hm][k]; }
      r2 = std::fmax(r2, dot3(d, d));
      for (int k = 0; k < 3; k++) if (kc.hand_v[i][k] < kc.hand_lo[k] - 1e-9 || kc.hand_v[i][k] > kc.hand_hi[k] + 1e-9) { *err = "hand hull outside its bounding box"; return -1; }
    }
    for (int k = 0; k < 3; k++) kc.hand_center[k] = m.mesh_center[hm][k];
    kc.hand_r = std::sqrt(r2);
    if (!(gb[SK_NB - 1] < gh && gh < ghull[0])) { *err = "unexpected hand geom order"; return -1; }
    if (mix(gb[0], gh, kc.set[SKS_BOXHAND])) { *err = "unsupported contact dimension"; return -1; }
    {   // translational body_invweight0 of the hand body at qpos0 (as for the finger bodies above; arm dofs only)
      double q[NDOF] = {0}, v[NDOF] = {0};
      DynOut dyn;
      dynamics(pcst, q, v, dyn);
      double L[45], d[NDOF], id[NDOF], Minv[NDOF][NDOF];
      ldl9(dyn.M, L, d, id);
      for (int col = 0; col < NDOF; col++) { double e[NDOF] = {0}; e[col] = 1; ldl9_solve(L, id, e); for (int r = 0; r < NDOF; r++) Minv[r][col] = e[r]; }
      double sn[NARM], cs[NARM], R7[9], p7[3], ax[NARM][3], og[NARM][3];
      for (int i = 0; i < NARM; i++) { sn[i] = 0; cs[i] = 1; }
      world_chain(pcst, sn, cs, R7, p7, ax, og);
      Xf xb = rel(link7, hand);
      double cl[3], cw[3];
      mv(xb.R, m.body_ipos[hand], cl); for (int k = 0; k < 3; k++) cl[k] += xb.p[k];
      mulE(R7, cl, cw); for (int k = 0; k < 3; k++) cw[k] += p7[k];
      double J[3][NDOF];
      for (int k = 0; k < NARM; k++) { double dd[3] = {cw[0] - og[k][0], cw[1] - og[k][1], cw[2] - og[k][2]}, col[3]; cross3(ax[k], dd, col); for (int r = 0; r < 3; r++) J[r][k] = col[r]; }
      for (int g = 0; g < NFING; g++) for (int r = 0; r < 3; r++) J[r][NARM + g] = 0.0;
      double tr = 0;
      for (int r = 0; r < 3; r++) for (int a = 0; a < NDOF; a++) for (int b = 0; b < NDOF; b++) tr += J[r][a] * Minv[a][b] * J[r][b];
      kc.invw_hand = std::fmax(1e-15, tr / 3);
    }
  }
  return 0;
}

}  // namespace d3il
