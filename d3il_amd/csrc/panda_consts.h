// panda_consts.h - compact, robot-specialised constant block consumed by the HIP kernels.
//
// Built once per handle on the host (build_panda_consts) from the generic model blob
// (include/d3il_model_blob.h) and uploaded to device memory; every lane of every wave reads it
// through uniform (scalar, SGPR) loads.  It is the "shared model" of the north star: body tree,
// joint limits, geom parameters, controller gains - ~3.5 KB.
//
// The reference keeps the same information in MjModel (mj_scene_parser.py:36-53), the pinocchio
// model (core/Model.py:26-35) and the gin-configured controller objects
// (controllers/GainsInterface.py:10-67).  Welded children are merged here (link7 + link8 + hand +
// tcp + rod + rod:tip; finger + fingertip), which MuJoCo does implicitly through its composite
// rigid body pass [ext].
#pragma once
#include <cmath>
#include <cstring>
#include "../../include/d3il_model_blob.h"

#if defined(__HIPCC__)
#define D3IL_HD __host__ __device__ __forceinline__
#else
#define D3IL_HD inline
#endif

namespace d3il {

constexpr int NARM = 7;      // revolute arm joints
constexpr int NFING = 2;     // prismatic finger joints
constexpr int NDOF = 9;
constexpr int MAXOBST = 8;

struct PandaConsts {
  // ---- dynamics chain (MJCF): link i frame = parent frame * (P[i], Q[i]) * Rz(q_i)
  double P[NARM][3];
  double Q[NARM][9];         // row-major rotation, parent <- link at q = 0
  double mass[NARM];
  double com[NARM][3];       // link frame
  double Ic[NARM][6];        // about the COM, link axes: xx yy zz xy xz yz
  // fingers: prismatic leaves on link 7 (all vectors in link-7 axes)
  double f_mass[NFING];
  double f_axis[NFING][3];
  double f_com0[NFING][3];   // COM at q = 0 relative to the link-7 origin
  double f_Ic[NFING][6];
  double f_damping[NFING];
  double gravity[3];
  double timestep;
  // joint limits (MJCF ranges) and their soft-constraint constants
  double jnt_range[NDOF][2];
  double lim_K[NDOF], lim_B[NDOF];      // spring/damper from solref, solimp[1]
  double lim_solimp[NDOF][5];
  double lim_margin[NDOF];
  double dof_invweight0[NDOF];
  double force_lo[NDOF], force_hi[NDOF];
  // points of link 7
  double tcp7[3];            // TCP origin in the link-7 frame
  double rod_c7[3], rod_u7[3], rod_r, rod_h;
  double rod_invweight0;     // body_invweight0[rod][0] (translational)
  // obstacles (static cylinders)
  int n_obst, pad_i;
  double ob_c[MAXOBST][3], ob_u[MAXOBST][3], ob_r[MAXOBST], ob_h[MAXOBST];
  double ct_K[MAXOBST], ct_B[MAXOBST], ct_solimp[MAXOBST][5], ct_margin[MAXOBST], ct_fric[MAXOBST][3];
  double impratio;
  // ---- controller kinematic chain (URDF): joint k placement (Kx, KR) then Rz(q_k); tool after joint 7
  double Kx[NARM][3];
  double KR[NARM][9];
  double tool_x[3], tool_R[9];
  // ---- controller gains
  double pd_p[NARM], pd_d[NARM];
  double ik_ppos[3], ik_pquat[3], ik_pnull[NARM], ik_rest[NARM], ik_W[NARM];
  double ik_Jreg, ik_filter, ik_minsv, ik_maxsv, ik_lr;
  double q_min[NARM], q_max[NARM];
  int ik_iters, n_substeps, max_steps, pad_j;
  // ---- task constants (avoiding.py:94-107)
  double task_f[16];
};

// ---------------------------------------------------------------- small host helpers
namespace hostmath {
inline void quat2mat(const double* q, double* M) {
  double w = q[0], x = q[1], y = q[2], z = q[3];
  M[0] = w * w + x * x - y * y - z * z; M[1] = 2 * (x * y - w * z); M[2] = 2 * (x * z + w * y);
  M[3] = 2 * (x * y + w * z); M[4] = w * w - x * x + y * y - z * z; M[5] = 2 * (y * z - w * x);
  M[6] = 2 * (x * z - w * y); M[7] = 2 * (y * z + w * x); M[8] = w * w - x * x - y * y + z * z;
}
inline void mm(const double* A, const double* B, double* C) {
  double t[9];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
  std::memcpy(C, t, sizeof t);
}
inline void mv(const double* A, const double* v, double* r) {
  double t[3] = {A[0] * v[0] + A[1] * v[1] + A[2] * v[2], A[3] * v[0] + A[4] * v[1] + A[5] * v[2], A[6] * v[0] + A[7] * v[1] + A[8] * v[2]};
  std::memcpy(r, t, sizeof t);
}
// rigid transform a <- b : x_a = R x_b + p
struct Xf { double R[9], p[3]; };
inline Xf identity() { Xf x; std::memset(&x, 0, sizeof x); x.R[0] = x.R[4] = x.R[8] = 1; return x; }
inline Xf compose(const Xf& a, const Xf& b) { Xf c; mm(a.R, b.R, c.R); mv(a.R, b.p, c.p); for (int k = 0; k < 3; k++) c.p[k] += a.p[k]; return c; }
// accumulate a rigid body (mass m, COM c, inertia tensor I about its COM; all in the target frame)
struct Acc { double m = 0, mc[3] = {0, 0, 0}, Io[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; };  // Io about the frame origin
inline void acc_add(Acc& a, double m, const double* c, const double* I) {
  a.m += m;
  double cc = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
  for (int i = 0; i < 3; i++) { a.mc[i] += m * c[i]; for (int j = 0; j < 3; j++) a.Io[3 * i + j] += I[3 * i + j] + m * ((i == j ? cc : 0) - c[i] * c[j]); }
}
inline void acc_finish(const Acc& a, double* mass, double* com, double* Ic6) {
  *mass = a.m;
  double c[3] = {a.mc[0] / a.m, a.mc[1] / a.m, a.mc[2] / a.m};
  double cc = c[0] * c[0] + c[1] * c[1] + c[2] * c[2], I[9];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) I[3 * i + j] = a.Io[3 * i + j] - a.m * ((i == j ? cc : 0) - c[i] * c[j]);
  std::memcpy(com, c, sizeof c);
  Ic6[0] = I[0]; Ic6[1] = I[4]; Ic6[2] = I[8]; Ic6[3] = I[1]; Ic6[4] = I[2]; Ic6[5] = I[5];
}
inline void solref_kb(const double* solref, const double* solimp, double h, double* K, double* B) {
  const double MINV = 1e-15;
  double dmax = std::fmin(0.9999, std::fmax(0.0001, solimp[1]));
  if (solref[0] > 0) {
    double tc = std::fmax(solref[0], 2 * h), dr = solref[1];
    *K = 1 / std::fmax(MINV, dmax * dmax * tc * tc * dr * dr);
    *B = 2 / std::fmax(MINV, dmax * tc);
  } else { *K = -solref[0] / std::fmax(MINV, dmax * dmax); *B = -solref[1] / std::fmax(MINV, dmax); }
}
inline void clamp_solimp(const double* in, double* out) {
  out[0] = std::fmin(0.9999, std::fmax(0.0001, in[0])); out[1] = std::fmin(0.9999, std::fmax(0.0001, in[1]));
  out[2] = std::fmax(0.0, in[2]); out[3] = std::fmin(0.9999, std::fmax(0.0001, in[3])); out[4] = std::fmax(1.0, in[4]);
}
}  // namespace hostmath

// Fills everything except dof_invweight0 / rod_invweight0 (those need the dynamics at qpos0 and are
// completed by finish_invweights() in panda_step.h's host section).  Returns 0 or a negative error.
inline int build_panda_consts(const d3il_model_blob& m, PandaConsts& c, const char** err) {
  using namespace hostmath;
  std::memset(&c, 0, sizeof c);
  if (m.magic != D3IL_BLOB_MAGIC || m.version != D3IL_BLOB_VERSION) { *err = "model blob: bad magic/version"; return -2; }
  if (m.nu != 9) { *err = "model blob: expected 9 actuators (7 arm + 2 fingers)"; return -3; }
  int jarm[NARM], jf[NFING];
  for (int k = 0; k < NARM; k++) jarm[k] = m.act_jnt[k];
  for (int k = 0; k < NFING; k++) jf[k] = m.act_jnt[NARM + k];
  // world <- body transform at q = 0 for every body (chain of body_pos/body_quat)
  static thread_local Xf X0[D3IL_MAXBODY];
  X0[0] = identity();
  for (int b = 1; b < m.nbody; b++) { Xf l; quat2mat(m.body_quat[b], l.R); std::memcpy(l.p, m.body_pos[b], sizeof l.p); X0[b] = compose(X0[m.body_parent[b]], l); }
  auto rel = [&](int a, int b) {  // a <- b at q = 0
    Xf inv; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) inv.R[3 * i + j] = X0[a].R[3 * j + i];
    double t[3]; mv(inv.R, X0[a].p, t); for (int k = 0; k < 3; k++) inv.p[k] = -t[k];
    return compose(inv, X0[b]);
  };
  auto weld_root = [&](int b) { while (b > 0 && m.body_jntnum[b] == 0) b = m.body_parent[b]; return b; };
  for (int k = 0; k < NARM; k++) {
    int j = jarm[k], b = m.jnt_body[j];
    if (m.jnt_type[j] != D3IL_JNT_HINGE || m.jnt_axis[j][2] != 1.0 || m.jnt_pos[j][0] != 0 || m.jnt_pos[j][1] != 0 || m.jnt_pos[j][2] != 0) { *err = "arm joints must be hinges about the link z axis through the link origin"; return -4; }
    int parent_link = k == 0 ? 0 : m.jnt_body[jarm[k - 1]];
    if (weld_root(m.body_parent[b]) != (k == 0 ? 0 : parent_link)) { *err = "arm is not a serial chain"; return -5; }
    Xf x = rel(parent_link, b);
    std::memcpy(c.P[k], x.p, sizeof x.p); std::memcpy(c.Q[k], x.R, sizeof x.R);
    Acc a;
    for (int bb = 1; bb < m.nbody; bb++) {
      if (weld_root(bb) != b || m.body_mass[bb] == 0) continue;
      Xf xb = rel(b, bb);
      double cw[3], Ri[9], Rt[9], D[9] = {m.body_inertia[bb][0], 0, 0, 0, m.body_inertia[bb][1], 0, 0, 0, m.body_inertia[bb][2]}, I[9], iq[9];
      mv(xb.R, m.body_ipos[bb], cw); for (int i = 0; i < 3; i++) cw[i] += xb.p[i];
      quat2mat(m.body_iquat[bb], iq); mm(xb.R, iq, Ri);
      for (int i = 0; i < 3; i++) for (int jj = 0; jj < 3; jj++) Rt[3 * i + jj] = Ri[3 * jj + i];
      mm(Ri, D, I); mm(I, Rt, I);
      acc_add(a, m.body_mass[bb], cw, I);
    }
    acc_finish(a, &c.mass[k], c.com[k], c.Ic[k]);
    c.jnt_range[k][0] = m.jnt_range[j][0]; c.jnt_range[k][1] = m.jnt_range[j][1];
  }
  int link7 = m.jnt_body[jarm[NARM - 1]];
  for (int k = 0; k < NFING; k++) {
    int j = jf[k], b = m.jnt_body[j];
    if (m.jnt_type[j] != D3IL_JNT_SLIDE || weld_root(m.body_parent[b]) != link7) { *err = "finger joints must be slides on the link-7 weld group"; return -6; }
    Xf x = rel(link7, b);
    mv(x.R, m.jnt_axis[j], c.f_axis[k]);
    Acc a;
    for (int bb = 1; bb < m.nbody; bb++) {
      if (weld_root(bb) != b || m.body_mass[bb] == 0) continue;
      Xf xb = rel(link7, bb);
      double cw[3], Ri[9], Rt[9], D[9] = {m.body_inertia[bb][0], 0, 0, 0, m.body_inertia[bb][1], 0, 0, 0, m.body_inertia[bb][2]}, I[9], iq[9];
      mv(xb.R, m.body_ipos[bb], cw); for (int i = 0; i < 3; i++) cw[i] += xb.p[i];
      quat2mat(m.body_iquat[bb], iq); mm(xb.R, iq, Ri);
      for (int i = 0; i < 3; i++) for (int jj = 0; jj < 3; jj++) Rt[3 * i + jj] = Ri[3 * jj + i];
      mm(Ri, D, I); mm(I, Rt, I);
      acc_add(a, m.body_mass[bb], cw, I);
    }
    acc_finish(a, &c.f_mass[k], c.f_com0[k], c.f_Ic[k]);
    c.f_damping[k] = m.jnt_damping[j];
    c.jnt_range[NARM + k][0] = m.jnt_range[j][0]; c.jnt_range[NARM + k][1] = m.jnt_range[j][1];
  }
  for (int k = 0; k < NDOF; k++) {
    int j = m.act_jnt[k];
    if (!m.jnt_limited[j]) { c.jnt_range[k][0] = -1e300; c.jnt_range[k][1] = 1e300; }
    solref_kb(m.jnt_solref[j], m.jnt_solimp[j], m.timestep, &c.lim_K[k], &c.lim_B[k]);
    clamp_solimp(m.jnt_solimp[j], c.lim_solimp[k]);
    c.lim_margin[k] = m.jnt_margin[j];
    c.force_lo[k] = m.act_forcelimited[k] ? m.act_forcerange[k][0] : -1e300;
    c.force_hi[k] = m.act_forcelimited[k] ? m.act_forcerange[k][1] : 1e300;
  }
  std::memcpy(c.gravity, m.gravity, sizeof c.gravity);
  c.timestep = m.timestep; c.impratio = m.impratio;
  { Xf x = rel(link7, m.tcp_body); std::memcpy(c.tcp7, x.p, sizeof x.p); }
  if (m.rod_geom >= 0) {
    int g = m.rod_geom, b = m.geom_body[g];
    if (weld_root(b) != link7 || m.geom_type[g] != D3IL_GEOM_CYLINDER) { *err = "rod must be a cylinder welded to link 7"; return -7; }
    Xf x = rel(link7, b); Xf gl; quat2mat(m.geom_quat[g], gl.R); std::memcpy(gl.p, m.geom_pos[g], sizeof gl.p);
    Xf xg = compose(x, gl);
    std::memcpy(c.rod_c7, xg.p, sizeof xg.p);
    c.rod_u7[0] = xg.R[2]; c.rod_u7[1] = xg.R[5]; c.rod_u7[2] = xg.R[8];
    c.rod_r = m.geom_size[g][0]; c.rod_h = m.geom_size[g][1];
  }
  c.n_obst = m.n_obst;
  for (int o = 0; o < m.n_obst; o++) {
    int g = m.obst_geom[o], b = m.geom_body[g];
    if (weld_root(b) != 0 || m.geom_type[g] != D3IL_GEOM_CYLINDER || m.rod_geom < 0) { *err = "obstacles must be static cylinders"; return -8; }
    Xf gl; quat2mat(m.geom_quat[g], gl.R); std::memcpy(gl.p, m.geom_pos[g], sizeof gl.p);
    Xf xg = compose(X0[b], gl);
    std::memcpy(c.ob_c[o], xg.p, sizeof xg.p);
    c.ob_u[o][0] = xg.R[2]; c.ob_u[o][1] = xg.R[5]; c.ob_u[o][2] = xg.R[8];
    c.ob_r[o] = m.geom_size[g][0]; c.ob_h[o] = m.geom_size[g][1];
    // contact parameter mixing, mj_contactParam [ext]: equal priority, solmix-weighted solref/solimp, max friction
    int r = m.rod_geom;
    if (m.geom_priority[g] != m.geom_priority[r]) { *err = "geom priority mixing not supported for rod contacts"; return -9; }
    double s1 = m.geom_solmix[g], s2 = m.geom_solmix[r], mix = s1 / (s1 + s2), solref[2], solimp[5];
    if (m.geom_solref[g][0] > 0 && m.geom_solref[r][0] > 0) for (int k = 0; k < 2; k++) solref[k] = mix * m.geom_solref[g][k] + (1 - mix) * m.geom_solref[r][k];
    else for (int k = 0; k < 2; k++) solref[k] = std::fmin(m.geom_solref[g][k], m.geom_solref[r][k]);
    for (int k = 0; k < 5; k++) solimp[k] = mix * m.geom_solimp[g][k] + (1 - mix) * m.geom_solimp[r][k];
    solref_kb(solref, solimp, m.timestep, &c.ct_K[o], &c.ct_B[o]);
    clamp_solimp(solimp, c.ct_solimp[o]);
    c.ct_margin[o] = std::fmax(m.geom_margin[g], m.geom_margin[r]) - std::fmax(m.geom_gap[g], m.geom_gap[r]);
    for (int k = 0; k < 3; k++) c.ct_fric[o][k] = std::fmax(m.geom_friction[g][k], m.geom_friction[r][k]);
    int cd = m.geom_condim[g] > m.geom_condim[r] ? m.geom_condim[g] : m.geom_condim[r];
    if (cd != 3) { *err = "rod contacts must have condim 3"; return -10; }
  }
  // controller chain (URDF): fold fixed joints into the next revolute placement / the tool
  {
    Xf acc = identity(); int k = 0;
    for (int i = 0; i < m.nchain; i++) {
      Xf l; std::memcpy(l.R, m.chain_R[i], sizeof l.R); std::memcpy(l.p, m.chain_xyz[i], sizeof l.p);
      acc = compose(acc, l);
      if (m.chain_type[i] == 1) {
        if (k >= NARM || m.chain_axis[i][0] != 0 || m.chain_axis[i][1] != 0 || m.chain_axis[i][2] != 1) { *err = "URDF chain: expected 7 revolute z-axis joints"; return -11; }
        std::memcpy(c.Kx[k], acc.p, sizeof acc.p); std::memcpy(c.KR[k], acc.R, sizeof acc.R);
        acc = identity(); k++;
      }
    }
    if (k != NARM) { *err = "URDF chain: expected 7 revolute joints"; return -11; }
    std::memcpy(c.tool_x, acc.p, sizeof acc.p); std::memcpy(c.tool_R, acc.R, sizeof acc.R);
  }
  for (int k = 0; k < NARM; k++) {
    c.pd_p[k] = m.pd_pgain[k]; c.pd_d[k] = m.pd_dgain[k]; c.ik_pnull[k] = m.ik_pgain_null[k]; c.ik_rest[k] = m.ik_rest[k];
    c.ik_W[k] = m.ik_W[k]; c.q_min[k] = m.ctrl_qmin[k]; c.q_max[k] = m.ctrl_qmax[k];
  }
  for (int k = 0; k < 3; k++) { c.ik_ppos[k] = m.ik_pgain_pos[k]; c.ik_pquat[k] = m.ik_pgain_quat[k]; }
  c.ik_Jreg = m.ik_J_reg; c.ik_filter = m.ik_filter; c.ik_minsv = m.ik_min_sv; c.ik_maxsv = m.ik_max_sv; c.ik_lr = m.ik_lr;
  c.ik_iters = m.ik_num_iter; c.n_substeps = m.n_substeps; c.max_steps = m.max_steps;
  for (int k = 0; k < 16; k++) c.task_f[k] = m.task_f[k];
  if (c.ik_filter != 1.0) { *err = "controller: joint_filter_coefficient != 1 not supported by the fused kernel"; return -12; }
  for (int k = 0; k < NARM; k++) if (c.ik_W[k] != 1.0) { *err = "controller: W != identity not supported by the fused kernel"; return -13; }
  return 0;
}

}  // namespace d3il
