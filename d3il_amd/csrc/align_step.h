// align_step.h - the Aligning task on the wave-cooperative engine of stack_step.h, variant SKV_ALIGNING.
//
// Reference path: simulation/aligning_sim.py:30-204 (rollout loop, the policy commands x, y AND z) -> Robot_Push_Env.step
// (gym_aligning/envs/aligning.py:282-286) over GymEnvWrapper.step (gyms/gym_env_wrapper.py:45-100) -> the Cartesian controller / mj_step chain
// of the rod robot, on a 15-dof model: the arm (9) and ONE free compound body (envs/.../robot_push_box.xml: a 10 x 10 x 2 cm plate of 1 kg with
// geom priority 1 and friction 0.3, carrying four 1 g walls).  The target body has sites only: its pose comes with the context and enters
// observation, reward and success (aligning.py:223-252, 314-342); behaviour mode 0 / 1 = rod within / beyond 5.1 cm of the box centre in xy
// (aligning.py:288-312).
//
// What this file holds: the constants of the engine variant from the model blob (build_coop_align_consts), the state layout that crosses the ABI
// (D3IL_ALIGN_STATE_* in include/d3il_rollout.h) and the task logic.  The physics is stack_step.h's: collision jobs per (geom, static) and
// (rod, geom) in lane groups, packed contact rows, two environments per wave in the Newton solver; the compound body is block 0 of the engine in
// centre-of-mass coordinates (see StackConsts).
#pragma once
#include "stack_step.h"

namespace d3il {

// arm q[9] v[9] bias[7] tcp[3] ik_q[7] ik_qd[7] | box pos3 quat4 vel6 | warm start: box qacc[6] (centre-of-mass coordinates) arm qacc[9] | target pos3 quat4
constexpr int AL_STATE_BOX = 42, AL_STATE_WARM = AL_STATE_BOX + 13, AL_STATE_TARGET = AL_STATE_WARM + 6 + NDOF, AL_STATE_F64 = AL_STATE_TARGET + 7;
constexpr int AL_OBS = 17, AL_CTX = 14;
struct AlignTask { double pos_min_dist, rot_min_dist, robot_box_dist; };
struct AlignState { EnvState arm; BoxState box; double target[7]; };

#if defined(__HIPCC__)
__constant__ AlignTask g_align_task;
#endif

// 2 arccos |p . q| (rotation_distance, aligning.py:22-31).  numpy returns NaN for |p . q| > 1 - reached by round-off when the two quaternions
// coincide -, and every comparison with that NaN is False: `nan` reports the case, the value is then unused (no arithmetic on a NaN: the device
// code is compiled with -ffinite-math-only)
D3IL_HD double align_rot_dist(const double* p, const double* q, bool* nan) {
  const double d = fabs(p[0] * q[0] + p[1] * q[1] + p[2] * q[2] + p[3] * q[3]);
  *nan = d > 1.0;
  return *nan ? 0.0 : 2.0 * acos(d);
}
D3IL_HD double align_dist3(const double* a, const double* b) { const double x = a[0] - b[0], y = a[1] - b[1], z = a[2] - b[2]; return sqrt(x * x + y * y + z * z); }
D3IL_HD double align_nan() {
#if defined(__HIP_DEVICE_COMPILE__)
  return __longlong_as_double(0x7ff8000000000000ll);
#else
  return std::nan("");
#endif
}
D3IL_HD void align_obs(const AlignState& as, float* obs) {   // get_observation, aligning.py:223-252: robot_pos | box pos, quat | target pos, quat
  for (int k = 0; k < 3; k++) obs[k] = (float)as.arm.tcp[k];
  for (int k = 0; k < 3; k++) obs[3 + k] = (float)as.box.pos[k];
  for (int k = 0; k < 4; k++) obs[6 + k] = (float)as.box.quat[k];
  for (int k = 0; k < 7; k++) obs[10 + k] = (float)as.target[k];
}
D3IL_HD bool align_success(const AlignTask& tk, const AlignState& as) {   // _check_early_termination, aligning.py:325-342
  bool nan;
  const double rot = align_rot_dist(as.box.quat, as.target + 3, &nan);
  return !nan && align_dist3(as.box.pos, as.target) <= tk.pos_min_dist && rot / 3.141592653589793 <= tk.rot_min_dist;
}
// before the physics of a step: observation, reward, done (gym_env_wrapper.py:88-90, 124-137)
D3IL_HD void align_step_begin(const AlignTask& tk, AlignState& as, float* obs, double* reward, unsigned char* done, int max_steps) {
  align_obs(as, obs);
  bool nan;
  const double rot = align_rot_dist(as.box.quat, as.target + 3, &nan);
  *reward = nan ? align_nan() : -rot / 3.141592653589793 + -3.5 * align_dist3(as.box.pos, as.target);      // get_reward, aligning.py:314-323
  bool fin = (as.arm.flags & F_TERMINATED) != 0;
  if (!fin && align_success(tk, as)) { as.arm.flags |= F_TERMINATED; fin = true; }
  if (!fin && as.arm.step >= max_steps - 1) fin = true;
  *done = fin ? 1 : 0;
}
// after the physics: success, behaviour mode, mean distance (aligning.py:282-312)
D3IL_HD void align_step_end(const AlignTask& tk, AlignState& as, double* mean_distance) {
  as.arm.step++;
  as.arm.flags &= ~F_SUCCESS;
  if (align_success(tk, as)) as.arm.flags |= F_SUCCESS | F_TERMINATED;
  const double dx = as.box.pos[0] - as.arm.tcp[0], dy = as.box.pos[1] - as.arm.tcp[1];
  const int mode = sqrt(dx * dx + dy * dy) < tk.robot_box_dist ? 0 : 1;
  as.arm.flags = (as.arm.flags & ~PF_MODE_MASK) | ((unsigned)(mode + 1) << PF_MODE_SHIFT);
  bool nan;
  const double rot = align_rot_dist(as.box.quat, as.target + 3, &nan);
  *mean_distance = nan ? align_nan() : 0.5 * (align_dist3(as.box.pos, as.target) + rot / 3.141592653589793);
}

// ------------------------------------------------------------------------------------------------ constants from the blob
D3IL_HOSTFN inline int build_coop_align_consts(const d3il_model_blob& m, const PandaConsts& pcst, StackConsts& kc, AlignTask& tk, const char** err) {
  std::memset(&kc, 0, sizeof kc);
  if (m.n_obj != 1) { *err = "aligning needs one task object"; return -1; }
  const int bd = m.obj_body[0];
  if (m.body_jntnum[bd] != 1 || m.jnt_type[m.body_jntadr[bd]] != D3IL_JNT_FREE) { *err = "the task object must be a free body"; return -1; }
  if (std::fabs(m.body_iquat[bd][0]) != 1.0) { *err = "the principal axes of the task object must be its body axes"; return -1; }
  kc.variant = SKV_ALIGNING; kc.nb = 1; kc.ns = 2;
  // block 0: the compound body about its centre of mass; blocks 1, 2: not part of this task (parked, inert)
  for (int b = 0; b < SK_NB; b++) {
    for (int k = 0; k < 3; k++) { kc.box_half[b][k] = 0.01; kc.box_inertia[b][k] = b == 0 ? m.body_inertia[bd][k] : 1.0; }
    kc.box_mass[b] = b == 0 ? m.body_mass[bd] : 1.0;
    kc.box_invw[b] = 1.0 / kc.box_mass[b];      // body_invweight0 is taken at the centre of mass (engine_setconst.c set0 [ext]): exactly 1 / m for a free body
    kc.box_r[b] = 0.02;
  }
  for (int k = 0; k < 3; k++) kc.al_c[k] = m.body_ipos[bd][k];
  int ng = 0, gg[AL_NG];
  for (int g = 0; g < m.ngeom; g++) {
    if (m.geom_body[g] != bd || !m.geom_contype[g]) continue;
    if (m.geom_type[g] != D3IL_GEOM_BOX || ng >= AL_NG) { *err = "the task object must consist of at most five box geoms"; return -1; }
    if (m.geom_quat[g][0] != 1.0 || m.geom_margin[g] != 0 || m.geom_gap[g] != 0 || m.geom_condim[g] != 3) { *err = "task-object geoms must be axis aligned, condim 3, without margin"; return -1; }
    if (g > m.rod_geom) { *err = "unexpected geom order (the task object's geoms must precede the rod)"; return -1; }
    gg[ng] = g;
    double r2 = 0, far2 = 0;
    for (int k = 0; k < 3; k++) {
      kc.al_gpos[ng][k] = m.geom_pos[g][k] - kc.al_c[k];      // relative to the centre of mass: the engine's "box position" of block 0
      kc.al_ghalf[ng][k] = m.geom_size[g][k];
      r2 += m.geom_size[g][k] * m.geom_size[g][k];
      const double f = std::fabs(m.geom_pos[g][k]) + m.geom_size[g][k]; far2 += f * f;
    }
    kc.al_gr[ng] = std::sqrt(r2);
    kc.al_r = std::fmax(kc.al_r, std::sqrt(far2));
    ng++;
  }
  if (ng < 1) { *err = "the task object has no collision geom"; return -1; }
  kc.al_ng = ng;
  // the two static slabs under the table top: table_plane (0.49 0.98 0.001) and support_body (0.49 0.98 0.4), lab_surrounding.xml:3-4,112-114
  auto slab = [&](double hx, double hy, double hz) {
    for (int g = 0; g < m.ngeom; g++) if (m.geom_type[g] == D3IL_GEOM_BOX && std::fabs(m.geom_size[g][0] - hx) < 1e-12 && std::fabs(m.geom_size[g][1] - hy) < 1e-12 && std::fabs(m.geom_size[g][2] - hz) < 1e-12) return g;
    return -1;
  };
  const int gs[2] = {slab(0.49, 0.98, 0.001), slab(0.49, 0.98, 0.4)};
  if (gs[0] < 0 || gs[1] < 0) { *err = "table slabs not found"; return -1; }
  for (int s = 0; s < 2; s++) {
    double p[3] = {m.geom_pos[gs[s]][0], m.geom_pos[gs[s]][1], m.geom_pos[gs[s]][2]};
    for (int b = m.geom_body[gs[s]]; b > 0; b = m.body_parent[b]) {
      if (m.body_quat[b][0] != 1.0 || m.body_jntnum[b] != 0) { *err = "slab must be static and axis aligned"; return -1; }
      for (int k = 0; k < 3; k++) p[k] += m.body_pos[b][k];
    }
    if (gs[s] > gg[0]) { *err = "unexpected geom order (the slabs must precede the task object)"; return -1; }
    for (int k = 0; k < 3; k++) { kc.st_c[s][k] = p[k]; kc.st_h[s][k] = m.geom_size[gs[s]][k]; }
    kc.st_R[s][0] = kc.st_R[s][4] = kc.st_R[s][8] = 1.0;
  }
  // contact parameters (mj_contactParam [ext]): the geom with the higher priority supplies solref / solimp / friction (the plate: priority 1,
  // friction 0.3), equal priorities mix (solmix 1: plain mean, friction = max)
  auto fill = [&](int g1, int g2, StackSet& ps) {
    const int src = m.geom_priority[g1] > m.geom_priority[g2] ? g1 : (m.geom_priority[g2] > m.geom_priority[g1] ? g2 : -1);
    double sr[2], si[5], fr;
    if (src >= 0) { for (int k = 0; k < 2; k++) sr[k] = m.geom_solref[src][k]; for (int k = 0; k < 5; k++) si[k] = m.geom_solimp[src][k]; fr = m.geom_friction[src][0]; }
    else {
      for (int k = 0; k < 2; k++) sr[k] = 0.5 * (m.geom_solref[g1][k] + m.geom_solref[g2][k]);
      for (int k = 0; k < 5; k++) si[k] = 0.5 * (m.geom_solimp[g1][k] + m.geom_solimp[g2][k]);
      fr = std::fmax(m.geom_friction[g1][0], m.geom_friction[g2][0]);
    }
    const double dmax = std::fmin(0.9999, std::fmax(0.0001, si[1])), tc = std::fmax(sr[0], 2 * m.timestep);
    ps.K = 1 / std::fmax(1e-15, dmax * dmax * tc * tc * sr[1] * sr[1]);
    ps.B = 2 / std::fmax(1e-15, dmax * tc);
    for (int k = 0; k < 5; k++) ps.solimp[k] = si[k];
    ps.solimp[0] = std::fmin(0.9999, std::fmax(0.0001, si[0])); ps.solimp[1] = dmax;
    ps.fric[0] = fr; ps.fric[1] = 0.005; ps.fric[2] = 0.0001;      // condim 3: only the sliding coefficient enters
    ps.margin = 0.0; ps.dim = 3;
  };
  if (m.rod_geom < 0) { *err = "no rod geom"; return -1; }
  // parameter sets: geom 0 (the plate) against static s -> SKS_STATIC + s; the walls share their parameters: against static s -> SKS_BOXBOX + s;
  // the rod against the plate -> SKS_BOXROD, against a wall -> SKS_BOXHAND (slots the Stacking / Pushing variants use for pairs this task does not have)
  for (int g = 0; g < ng; g++) {
    if (g >= 2) for (int k = 0; k < 3; k++) if (m.geom_friction[gg[g]][k] != m.geom_friction[gg[1]][k] || m.geom_priority[gg[g]] != m.geom_priority[gg[1]]) { *err = "the walls must share their contact parameters"; return -1; }
    kc.al_set_static[g] = g == 0 ? SKS_STATIC : SKS_BOXBOX;
    kc.al_set_rod[g] = g == 0 ? SKS_BOXROD : SKS_BOXHAND;
  }
  for (int s = 0; s < 2; s++) { fill(gs[s], gg[0], kc.set[SKS_STATIC + s]); if (ng > 1) fill(gs[s], gg[1], kc.set[SKS_BOXBOX + s]); }
  fill(gg[0], m.rod_geom, kc.set[SKS_BOXROD]);
  if (ng > 1) fill(gg[1], m.rod_geom, kc.set[SKS_BOXHAND]);
  kc.impratio = m.impratio;
  // a body centre beyond the table top (less the body's bounding radius) raises OFF_TABLE: the frame beams around the table are not modelled here
  for (int k = 0; k < 2; k++) { kc.ws_lo[k] = kc.st_c[0][k] - (kc.st_h[0][k] - kc.al_r); kc.ws_hi[k] = kc.st_c[0][k] + (kc.st_h[0][k] - kc.al_r); }
  for (int k = 0; k < 3; k++) { kc.rod_c7[k] = pcst.rod_c7[k]; kc.rod_u7[k] = pcst.rod_u7[k]; }
  kc.rod_r = pcst.rod_r; kc.rod_h = pcst.rod_h; kc.invw_rod = pcst.rod_invweight0;
  tk.pos_min_dist = m.task_f[0]; tk.rot_min_dist = m.task_f[1]; tk.robot_box_dist = m.task_f[2];
  return 0;
}

}  // namespace d3il
