// hostcheck.cpp - TEST-ONLY host build of the kernels' per-environment math (d3il_amd/csrc/panda_step.h).
// Lets the CPU test-suite compare the exact device formulation against the oracle without a GPU.
// It is NOT part of the product: libd3il_rollout.so never links or calls this file, and the product
// fails loudly when no HIP device is present.
#define D3IL_HOST_STATS 1
#include <cmath>
#include <cstdlib>
#include <cstring>
#include "../../d3il_amd/csrc/gen_step.h"
#include "../../d3il_amd/csrc/stack_step.h"

using namespace d3il;

static void unpack(const double* s, const int* f, EnvState& st) {
  int k = 0;
  for (int i = 0; i < NDOF; i++) st.q[i] = s[k++];
  for (int i = 0; i < NDOF; i++) st.v[i] = s[k++];
  for (int i = 0; i < NARM; i++) st.bias[i] = s[k++];
  for (int i = 0; i < 3; i++) st.tcp[i] = s[k++];
  for (int i = 0; i < NARM; i++) st.ikq[i] = s[k++];
  for (int i = 0; i < NARM; i++) st.ikqd[i] = s[k++];
  st.flags = (unsigned)f[0]; st.step = f[1];
}
static void pack(const EnvState& st, double* s, int* f) {
  int k = 0;
  for (int i = 0; i < NDOF; i++) s[k++] = st.q[i];
  for (int i = 0; i < NDOF; i++) s[k++] = st.v[i];
  for (int i = 0; i < NARM; i++) s[k++] = st.bias[i];
  for (int i = 0; i < 3; i++) s[k++] = st.tcp[i];
  for (int i = 0; i < NARM; i++) s[k++] = st.ikq[i];
  for (int i = 0; i < NARM; i++) s[k++] = st.ikqd[i];
  f[0] = (int)st.flags; f[1] = st.step;
}

extern "C" {
void* hc_create(const d3il_model_blob* blob, const char** err) {
  PandaConsts* c = (PandaConsts*)std::malloc(sizeof(PandaConsts));
  static const char* e = "";
  int rc = build_panda_consts(*blob, *c, &e);
  if (rc) { *err = e; std::free(c); return nullptr; }
  finish_invweights(*c);
  return c;
}
void hc_destroy(void* c) { std::free(c); }
void hc_stats(long* out, int reset) { out[0] = g_stats.newton_calls; out[1] = g_stats.newton_iters; out[2] = g_stats.ls_iters; out[3] = g_stats.eig_calls; out[4] = g_stats.ik_calls; out[5] = g_stats.contact_calls; if (reset) g_stats = Stats{0, 0, 0, 0, 0, 0}; }
// 0: the production stopping rule of the contact solvers, 1: the oracle's (iterate to round-off) - the device's option solver_strict
void hc_set_solver_strict(int strict) { host_solver_tol() = strict ? SOLVER_TOL_STRICT : SOLVER_TOL_PRODUCTION; }
int hc_sizeof_consts() { return (int)sizeof(PandaConsts); }
void hc_get_consts(const void* c, double* dof_invw, double* rod_invw, double* masses, double* coms) {
  const PandaConsts& p = *(const PandaConsts*)c;
  std::memcpy(dof_invw, p.dof_invweight0, sizeof p.dof_invweight0); *rod_invw = p.rod_invweight0;
  std::memcpy(masses, p.mass, sizeof p.mass); std::memcpy(coms, p.com, sizeof p.com);
}
void hc_dynamics(const void* c, const double* q, const double* v, double* M81, double* bias, double* tcp) {
  const PandaConsts& p = *(const PandaConsts*)c;
  DynOut d; dynamics(p, q, v, d);
  for (int r = 0; r < NDOF; r++) for (int k = 0; k < NDOF; k++) M81[r * NDOF + k] = d.M[r >= k ? tri(r, k) : tri(k, r)];
  std::memcpy(bias, d.bias, sizeof d.bias);
  double t[3]; mulE(d.R7, p.tcp7, t); for (int k = 0; k < 3; k++) tcp[k] = d.p7[k] + t[k];
}
void hc_ik_fk(const void* c, const double* q, double* pos, double* quat, double* J42) {
  const PandaConsts& p = *(const PandaConsts*)c;
  double R[9], ax[NARM][3], og[NARM][3], sq[NARM], cq[NARM];
  for (int k = 0; k < NARM; k++) { sq[k] = std::sin(q[k]); cq[k] = std::cos(q[k]); }
  ik_chain(p, sq, cq, pos, R, ax, og); mat2quat(R, quat);
  for (int k = 0; k < NARM; k++) { double d[3] = {pos[0] - og[k][0], pos[1] - og[k][1], pos[2] - og[k][2]}, cr[3]; cross3(ax[k], d, cr);
    for (int r = 0; r < 3; r++) { J42[r * NARM + k] = cr[r]; J42[(3 + r) * NARM + k] = ax[k][r]; } }
}
// one controller call: returns PD torque (without gravity compensation), updates ikq/ikqd/flags
void hc_ik_control(const void* c, const double* setpoint, const double* cur_q, const double* cur_v, double* ikq, double* ikqd, int* flags, int fast, double* tau) {
  const PandaConsts& p = *(const PandaConsts*)c;
  unsigned f = (unsigned)*flags;
  double n = std::sqrt(setpoint[3] * setpoint[3] + setpoint[4] * setpoint[4] + setpoint[5] * setpoint[5] + setpoint[6] * setpoint[6]);
  double dq[4] = {setpoint[3] / n, setpoint[4] / n, setpoint[5] / n, setpoint[6] / n};
  if (fast) ik_update<true>(p, setpoint, dq, cur_q, f, ikq, ikqd); else ik_update<false>(p, setpoint, dq, cur_q, f, ikq, ikqd);
  for (int k = 0; k < NARM; k++) tau[k] = p.pd_p[k] * (ikq[k] - cur_q[k]) + p.pd_d[k] * (ikqd[k] - cur_v[k]);
  *flags = (int)f;
}
void hc_solve6(const double* A21, const double* b, double lo, double hi, double* x, int fast) { if (fast) ik_solve6<true>(A21, b, lo, hi, x); else ik_solve6<false>(A21, b, lo, hi, x); }
void hc_physics_substep(const void* c, double* s, int* f, const double* tau, const double* ffing) {
  EnvState st; unpack(s, f, st); double warm[6]; warm[5] = 0; physics_substep(*(const PandaConsts*)c, st, tau, ffing, warm); pack(st, s, f);
}
void hc_env_reset(const void* c, const double* init_qpos, double* s, int* f, float* obs) {
  EnvState st; std::memset(&st, 0, sizeof st);
  env_reset(*(const PandaConsts*)c, st, init_qpos, obs); pack(st, s, f);
}
void hc_env_step(const void* c, double* s, int* f, const double* action, float* obs, unsigned char* done, int fast) {
  EnvState st; unpack(s, f, st);
  const PandaConsts& pc = *(const PandaConsts*)c;
  if (fast) env_step<true>(pc, st, action, obs, done, pc.n_substeps, pc.max_steps); else env_step<false>(pc, st, action, obs, done, pc.n_substeps, pc.max_steps);
  pack(st, s, f);
}

// ---------------------------------------------------------------- generic engine / Sorting (gen_step.h)
struct GenHost { PandaConsts c; GenConsts gc; double h[GL_SIZE]; double g[GG_SIZE]; };
void* hc_gen_create(const d3il_model_blob* blob, const char** err) {
  GenHost* p = (GenHost*)std::calloc(1, sizeof(GenHost));
  static const char* e = "";
  if (build_panda_consts(*blob, p->c, &e)) { *err = e; std::free(p); return nullptr; }
  finish_invweights(p->c);
  if (build_gen_consts(*blob, p->c, p->gc, &e)) { *err = e; std::free(p); return nullptr; }
  return p;
}
int hc_gen_info(void* h, int* nb, int* ns, double* statics /* ns x (c3, h3, first) */) {
  GenHost* p = (GenHost*)h; *nb = p->gc.nb; *ns = p->gc.ns;
  for (int s = 0; s < p->gc.ns; s++) { for (int k = 0; k < 3; k++) { statics[7 * s + k] = p->gc.st_c[s][k]; statics[7 * s + 3 + k] = p->gc.st_h[s][k]; } statics[7 * s + 6] = p->gc.st_first[s]; }
  return gen_state_rows(p->gc.nb);
}
// the split of the static list: [0, ns_core) inside the table, the rest = frame beams; and the two workspaces (inner lo / hi, outer lo / hi)
int hc_gen_workspace(void* h, double* ws8) {
  GenHost* p = (GenHost*)h;
  for (int k = 0; k < 2; k++) { ws8[k] = p->gc.in_lo[k]; ws8[2 + k] = p->gc.in_hi[k]; ws8[4 + k] = p->gc.ws_lo[k]; ws8[6 + k] = p->gc.ws_hi[k]; }
  return p->gc.ns_core;
}
// s: the environment's state column (arm[42] | cubes | warm start | task words), f: flags, step
void hc_gen_reset(void* h, const double* init_qpos, const double* ctx, double* s, int* f, float* obs) {
  GenHost* p = (GenHost*)h; EnvState st; std::memset(&st, 0, sizeof st);
  PushScratch sc{p->h, p->g, 1, s + 42, 1};
  gen_env_reset(p->c, p->gc, st, sc, init_qpos, ctx, obs); pack(st, s, f);
}
void hc_gen_step(void* h, double* s, int* f, const double* action, float* obs, unsigned char* done, int* mode_code, int fast) {
  GenHost* p = (GenHost*)h; EnvState st; unpack(s, f, st);
  PushScratch sc{p->h, p->g, 1, s + 42, 1};
  if (fast) gen_env_step<true>(p->c, p->gc, st, sc, action, obs, done, mode_code, p->c.n_substeps, p->c.max_steps);
  else gen_env_step<false>(p->c, p->gc, st, sc, action, obs, done, mode_code, p->c.n_substeps, p->c.max_steps);
  pack(st, s, f);
}
void hc_gen_island_hist(long* out, int reset) { for (int k = 0; k < 80; k++) { out[k] = g_isl_hist[k]; if (reset) g_isl_hist[k] = 0; } }
void hc_gen_substep(void* h, double* s, int* f, const double* tau, const double* ffing) {
  GenHost* p = (GenHost*)h; EnvState st; unpack(s, f, st);
  PushScratch sc{p->h, p->g, 1, s + 42, 1};
  gen_physics_substep(p->c, p->gc, st, sc, tau, ffing); pack(st, s, f);
}
// ---------------------------------------------------------------- Stacking (stack_step.h)
struct StackHost { PandaConsts c; StackConsts kc; double t[ST_SIZE]; double g[SG_SIZE]; };
static void stack_unpack(const double* s, const int* f, StackState& ss) {
  int k = 0;
  for (int i = 0; i < NDOF; i++) ss.arm.q[i] = s[k++];
  for (int i = 0; i < NDOF; i++) ss.arm.v[i] = s[k++];
  for (int i = 0; i < NARM; i++) ss.arm.bias[i] = s[k++];
  for (int i = 0; i < 3; i++) ss.arm.tcp[i] = s[k++];
  for (int b = 0; b < SK_NB; b++) { for (int i = 0; i < 3; i++) ss.box[b].pos[i] = s[k++]; for (int i = 0; i < 4; i++) ss.box[b].quat[i] = s[k++]; for (int i = 0; i < 6; i++) ss.box[b].vel[i] = s[k++]; }
  for (int i = 0; i < SK_NV; i++) ss.warm[i] = s[k++];
  ss.arm.flags = (unsigned)f[0]; ss.arm.step = f[1];
}
static void stack_pack(const StackState& ss, double* s, int* f) {
  int k = 0;
  for (int i = 0; i < NDOF; i++) s[k++] = ss.arm.q[i];
  for (int i = 0; i < NDOF; i++) s[k++] = ss.arm.v[i];
  for (int i = 0; i < NARM; i++) s[k++] = ss.arm.bias[i];
  for (int i = 0; i < 3; i++) s[k++] = ss.arm.tcp[i];
  for (int b = 0; b < SK_NB; b++) { for (int i = 0; i < 3; i++) s[k++] = ss.box[b].pos[i]; for (int i = 0; i < 4; i++) s[k++] = ss.box[b].quat[i]; for (int i = 0; i < 6; i++) s[k++] = ss.box[b].vel[i]; }
  for (int i = 0; i < SK_NV; i++) s[k++] = ss.warm[i];
  f[0] = (int)ss.arm.flags; f[1] = ss.arm.step;
}
void* hc_stack_create(const d3il_model_blob* blob, const char** err) {
  StackHost* p = (StackHost*)std::calloc(1, sizeof(StackHost));
  static const char* e = "";
  if (build_panda_consts(*blob, p->c, &e)) { *err = e; std::free(p); return nullptr; }
  finish_invweights(p->c);
  if (build_stack_consts(*blob, p->c, p->kc, &e)) { *err = e; std::free(p); return nullptr; }
  return p;
}
int hc_stack_state_size() { return SK_STATE_F64; }
void hc_stack_reset(void* h, const double* init_qpos, const double* ctx, double* s, int* f, float* obs) {
  StackHost* p = (StackHost*)h; StackState ss; std::memset(&ss, 0, sizeof ss);
  StackScratch sc{p->t, p->g};
  stack_env_reset(p->c, p->kc, ss, sc, init_qpos, ctx, obs); stack_pack(ss, s, f);
}
static int g_stack_poison = 0;
void hc_stack_poison(int on) { g_stack_poison = on; }
void hc_stack_step(void* h, double* s, int* f, const double* action, float* obs, unsigned char* done, double* mean_dist) {
  StackHost* p = (StackHost*)h; StackState ss; stack_unpack(s, f, ss);
  if (g_stack_poison) {   // read-before-write detector: the scratch areas hold NaN before every step
    for (int i = 0; i < ST_SIZE; i++) p->t[i] = std::nan("");
    for (int i = 0; i < SG_SIZE; i++) p->g[i] = std::nan("");
  }
  StackScratch sc{p->t, p->g};
  stack_env_step(p->c, p->kc, ss, sc, action, obs, done, mean_dist, p->c.n_substeps, p->c.max_steps);
  stack_pack(ss, s, f);
}
void hc_stack_scratch(void* h, double* out, int count) { StackHost* p = (StackHost*)h; std::memcpy(out, p->g, (size_t)count * sizeof(double)); }
// contacts of the last sub-step: per contact dist, pos3, normal3, bodyA, bodyB, set
int hc_stack_contacts(void* h, double* out) {
  StackHost* p = (StackHost*)h; int n = 0;
  for (int ci = 0; ci < SK_MAXCON; ci++) {
    const double* r = p->g + ci * SREC;
    if (r[3] == 0 && r[4] == 0 && r[5] == 0) break;
    out[10 * n] = r[12]; for (int k = 0; k < 3; k++) { out[10 * n + 1 + k] = r[k]; out[10 * n + 4 + k] = r[3 + k]; }
    out[10 * n + 7] = r[13]; out[10 * n + 8] = r[14]; out[10 * n + 9] = r[15]; n++;
  }
  return n;
}
// collision routines of push_step.h on their own (quaternions in, same record layout as the oracle's test hooks)
int hc_cyl_box(const double* pc, const double* qc, double rad, double half, const double* pb, const double* qb, const double* sb, double margin, double* out) {
  double Rc[9], Rb[9]; quat2mat(qc, Rc); quat2mat(qb, Rb);
  double axis[3] = {Rc[2], Rc[5], Rc[8]};
  return cyl_box(pc, axis, rad, half, pb, Rb, sb, margin, out) ? 1 : 0;
}
int hc_box_box(const double* p1, const double* q1, const double* s1, const double* p2, const double* q2, const double* s2, double margin, double* out) {
  double R1[9], R2[9], rec[16][7]; quat2mat(q1, R1); quat2mat(q2, R2);
  int n = box_box(p1, R1, s1, p2, R2, s2, margin, rec, 16);
  std::memcpy(out, rec, n * 7 * sizeof(double));
  return n;
}
}
