// gen_step.h - generic sub-step for "Panda arm + rod + NB free cubes + NS static boxes": the engine of the Sorting task
// (4 cubes, 2 table slabs, 8 bin walls, platform).
//
// An environment is worked on by a group of NB lanes, lane l owning cube l (lane 0 also owns the arm):
//   phase 1  lane 0    arm forward dynamics, smooth acceleration, limit rows, rod pose; the arm-alone solution (finger limit
//                      rows in closed form) as the default arm result
//   phase 2  lane l    cube l: rotation matrix, collision against the static boxes (sphere pre-test, then box_box in the
//                      model's geom order)
//   phase 3  lane l    cube l against the cubes after it and against the rod; lane 0 then adds the arm Jacobian rows of
//                      the rod contacts
//   phase 4            the constraint system is split into islands (connected components of cubes and arm under cube-cube
//                      and rod contacts); primal Newton with safeguarded exact line search, elliptic cones (condim 3), contact
//                      parameters mixed per pair (priority: the platform's own solref / solimp / friction):
//            lane l    a cube on its own - resting on the platform or the table: 6 dofs, 4 .. 8 contacts - is solved by its
//                      lane in registers (gen_solve_cube);
//            all lanes a larger island (the arm joins one only through a rod contact) is solved by the lanes of the group
//                      together (gen_solve): contacts dealt out per lane, compact dense system in LDS, blocked Cholesky
//   phase 5  lane l    semi-implicit Euler: arm with implicit finger damping (lane 0), cube l with quaternion integration
// The group's lanes are lanes of one wavefront on the device (wave-level fences between the phases, gen_kernels.h); the host
// build runs the lanes one after the other.  Per-environment working set: vectors, rotation matrices, arm mass matrix,
// limit rows and the dense (6 NB + 9)^2 Hessian in LDS (t area); contact records in HBM (g area); the cubes' state in the
// state buffer itself (w area).  Same numerics as the oracle's generic engine up to solver tolerance
// (tests/test_sorting_host.py, tests/test_gpu_parity_sorting.py).
#pragma once
#include "rigid_common.h"

namespace d3il {

constexpr int GEN_MAXNB = 4, GEN_MAXNS = 28, GEN_SEG = 24;
constexpr int GEN_ARMSEG = GEN_MAXNB;                // segment of the contacts that involve no cube (rod <-> static box); GEN_ARMCON of them take part in a solve
constexpr int GEN_ARMCON = 3, GEN_ARMLANE = 2;       // GEN_ARMLANE: what one lane of the group may find (it scans every nl-th static box)
constexpr int GEN_MAXCON = (GEN_MAXNB + 1) * GEN_SEG;
static_assert(GEN_ARMLANE * GEN_MAXNB <= GEN_SEG && GEN_ARMCON <= GEN_SEG, "the lanes' parking slots and the packed rod <-> static contacts share one record segment");
constexpr int GEN_MAXSET = 2 * GEN_MAXNS + 2;        // static s <-> cube: s;  cube <-> cube: ns;  rod <-> cube: ns + 1;  rod <-> static s: ns + 2 + s
enum { GEN_TASK_SORTING = 0, GEN_TASK_INSERTING = 1, GEN_TASK_PUSHING = 2 };
constexpr int GEN_MAXNV = 6 * GEN_MAXNB + NDOF;     // 33
constexpr int GEN_NH = GEN_MAXNV * (GEN_MAXNV + 1) / 2;   // 561
#ifndef D3IL_GEN_LANES
#define D3IL_GEN_LANES 16
#endif
constexpr int GEN_LANES = D3IL_GEN_LANES;     // environments per workgroup; a power of two
// Sub-lanes: on the device a cube is owned by a PAIR of neighbouring lanes that deal the cube's contacts out between them in the solver's contact
// loops and exchange their partial sums with a DPP swap (gen_tree.h; the joint solver gen_solve simply sees twice the lanes).  GEN_LANES / GEN_NSUB
// environments x GEN_MAXNB cubes x GEN_NSUB sub-lanes are one wavefront; a workgroup runs GEN_NSUB such physics waves.  The host build has one.
#if !defined(D3IL_GEN_NSUB)
#if defined(__HIP_DEVICE_COMPILE__) || (defined(__HIPCC__) && !defined(D3IL_HOST_ONLY))
#define D3IL_GEN_NSUB 2
#else
#define D3IL_GEN_NSUB 1
#endif
#endif
constexpr int GEN_NSUB = D3IL_GEN_NSUB;
constexpr int GEN_WAVE = 64;

struct GenConsts {
  int nb, ns, set_bb, set_rod;
  int ns_core;                    // statics [0, ns_core): everything inside the table; [ns_core, ns): the frame beams around the table edge (only tested near it)
  int task;                       // GEN_TASK_SORTING / GEN_TASK_INSERTING / GEN_TASK_PUSHING: which task logic reads the cubes (sort_* / ins_* / gpush_* below)
  int rod_static;                 // 1: the rod <-> static box pairs are evaluated (Inserting: the rod works between the walls of the gates)
  int st_rod[GEN_MAXNS];          // the rod's collision bits match static s
  double ins_target[9], ins_min_dist;      // Inserting: the three goal positions and target_min_dist (gate_insertion_objects.py:17-24, gate_insertion.py:276); Pushing: the red and the green target, min_dist (pushing_objects.py:11-15, pushing.py:251)
  double box_half[3], box_mass, box_inertia, box_invw_t;
  double st_c[GEN_MAXNS][3], st_h[GEN_MAXNS][3], st_R[GEN_MAXNS][9];
  int st_first[GEN_MAXNS];        // 1: the static geom precedes the cube geoms in the model (it is geom 1 of the pair)
  double ct_K[GEN_MAXSET], ct_B[GEN_MAXSET], ct_solimp[GEN_MAXSET][5], ct_fric[GEN_MAXSET];   // set s < ns: static s <-> cube
  double impratio, rod_invw;
  double ws_lo[2], ws_hi[2];      // modelled workspace of the cube centres (x, y): the table with its frame; a centre beyond it raises PF_OFF_TABLE
  double in_lo[2], in_hi[2];      // the part of it in which no frame beam can be reached
  double absent[7];               // pose reported for boxes the model does not have (body id -1)
};

// On the device the constants sit in constant memory (uniform reads become scalar loads); every function re-binds its `gc`
// to that object so that no generic pointer to it survives a call boundary.  One Sorting model per process.
#if defined(__HIPCC__)
__constant__ GenConsts g_gen_consts;
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define D3IL_GEN_CONSTS(in, name) const GenConsts& name = g_gen_consts; (void)in
#else
#define D3IL_GEN_CONSTS(in, name) const GenConsts& name = in
#endif

// The w area of the scratch views is the environment's object block of the state buffer (rows 42 ..): the cubes' pos[3] quat[4]
// vel[6], then the solver's warm start [nv], then the two task words (stored as doubles).  The cubes stay there for the
// whole step - only the arm lives in registers.
// g area (contact records, HBM): on the device the 16 environments of a workgroup own one block, [record field][environment column] - a record's fields
// are 128 bytes apart, so one 64-bit base address per record and immediate offsets address them (a row stride of n_envs x 8 bytes cost two
// address instructions per field) and an environment's records share their cache lines with its workgroup only.  Host build: the caller's stride.
#if defined(__HIP_DEVICE_COMPILE__)
// fields 2 k and 2 k + 1 of a record sit next to each other, [field pair][environment column][2]: the sixteen lanes that own eight environments in a
// physics wave (two sub-lanes each) store one field PAIR with one instruction - a full 128-byte line (gen_put).  Half-line stores (one field, eight
// environments) are written through to memory by this L2: 13 x the write traffic (profiles/r05/README.md).
#define GRS(i) sc.g[((i) >> 1) * (2 * GEN_LANES) + ((i) & 1)]
#else
#define GRS(i) sc.g[(long)(i) * sc.gs]
#endif
#define GBX(b, k) PWS(13 * (b) + (k))
#define GWARM(k) PWS(13 * gc.nb + (k))
#define GTASK(k) PWS(13 * gc.nb + 6 * gc.nb + NDOF + (k))
D3IL_HD constexpr int gen_state_rows(int nb) { return 42 + 13 * nb + 6 * nb + NDOF + 2; }

// t area (LDS): one CONTIGUOUS block of GL_SIZE doubles per environment, sc.h = its first word.  A field with a constant index is then one DS instruction with an
// immediate offset (< 64 KB) on the environment's base register.  Rounds 3 - 5 kept the area field-major ([field][environment column], stride 17 doubles): every
// field beyond 64 KB / 136 B needed an address register of its own, the optimiser hoisted those out of the sub-step loop by the hundred and the register allocator
// put them into scratch - 400 scratch loads per wave and sub-step (round 6, profiles/r06/README.md).  Bank spread: GL_SIZE is ODD, so the blocks of the 16
// environments of a workgroup start on 16 different bank pairs (a 64-bit access takes two of the 64 banks): the same field of different environments never collides.
#define GLS(i) sc.h[(i)]
constexpr int GL_H = 0, GL_X = GEN_NH, GL_P = GL_X + GEN_MAXNV, GL_G = GL_P + GEN_MAXNV, GL_A0 = GL_G + GEN_MAXNV, GL_VEL = GL_A0 + GEN_MAXNV;
constexpr int GL_R = GL_VEL + GEN_MAXNV, GL_POS = GL_R + 9 * GEN_MAXNB, GL_M = GL_POS + 3 * GEN_MAXNB, GL_LIM = GL_M + 45, GL_JA = GL_LIM + 27;
constexpr int GL_ROD = GL_JA + 21 * (GEN_MAXNB + GEN_ARMCON);      // rod centre[3], axis[3].  GL_JA: arm rows of the rod contact of cube b at 21 b, of rod <-> static contact j at 21 (GEN_MAXNB + j)
constexpr int GL_INFO = GL_ROD + 6;                 // [0..3] per cube: contact count | partner cubes << 5 | rod contact << 9;  [4] arm joint at a limit;  [5..7] flags of lanes 1..3;
                                                    // [8] rod <-> static contacts of this sub-step;  [9..12] what lanes 0..3 found of them
constexpr int GL_RED = GL_INFO + 16;                 // line-search partial sums of the group's lanes, double buffered: 2 x (4 GEN_NSUB) x (d1, d2)
constexpr int GL_NRED = 2 * GEN_MAXNB * GEN_NSUB;    // one buffer
constexpr int GL_TR = GL_RED + 2 * GL_NRED;                  // tree solver (gen_tree.h): [0] 1 = the arm's reduction to the lambda node stands, 2 = the arm's island was solved through it; [1..5] lambda
constexpr int GL_PAIR = GL_TR + 6;                  // cube pair (c, d), c < d: first record | count << 5 of their contacts in c's segment
constexpr int GL_ARMST = GL_PAIR + 6;               // the arm's state between the phases of the step kernel (q[9] v[9] bias[7] tcp[3] flags step: gen_kernels.h gen_park_arm) - NOT held in registers across the phases
constexpr int GL_SIZE = GL_ARMST + 30;
static_assert(GL_SIZE % 2 == 1, "odd block size: the environments' blocks start on different LDS bank pairs");
// g area (HBM): contact records, GEN_SEG per cube
constexpr int GG_CON = 0;
constexpr int GREC = 28;   // pos[3] frame[9] dist kind a b | aref[3] Dn fric sign | jar[3] jp[3]   (sign: of the segment's cube in the row, +1 = it is geom 2)
constexpr int GG_SIZE = GG_CON + GEN_MAXCON * GREC;
// rows of a workgroup's block: an ODD number of 128-byte rows, so that the same record field of neighbouring workgroups does not fall on the same L2
// channel / set (with 3360 rows = 105 x 4 KiB every workgroup's hot lines shared their low address bits: 30 x the write-backs, profiles/r05/README.md)
constexpr int GG_BLOCK = GG_SIZE + 1;
D3IL_HD int gt_pair(int c, int d) { return c * (2 * GEN_MAXNB - c - 1) / 2 + (d - c - 1); }      // slot of the cube pair c < d in GL_PAIR
enum { GK_STATIC = 0, GK_BOXBOX = 1, GK_ROD = 2, GK_RODST = 3 /* rod <-> static box: a = static, b = its slot of the GL_JA rows; no cube */ };

D3IL_HD void gen_sync() {     // orders the LDS / HBM traffic of the lanes of a group between two phases
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
#endif
}

// ---- islands.  mask: bit b = cube b, bit nb = arm.  An island's blocks sit in a packed list (4 bits per entry, ascending:
// cubes, then the arm), and its Newton system is stored compactly: block k of the list owns the compact dofs 6 k .. (the arm,
// always last, 9 of them).  The gradient / direction vectors (voff) and the packed lower Hessian (hoff) of the islands of an
// environment are laid out one after the other in the t area, so the lanes of a group can solve their islands side by side.
struct Isl { unsigned list; int n, m, hoff, voff; bool arm; };
D3IL_HD Isl gen_island(unsigned mask, int nb) {
  Isl s{0u, 0, 0, 0, 0, false};
  for (int b = 0; b <= nb; b++) if ((mask >> b) & 1) { s.list |= (unsigned)b << (4 * s.n); s.n++; s.m += b < nb ? 6 : NDOF; }
  s.arm = ((mask >> nb) & 1) != 0;
  return s;
}
#define ISL_BLK(k) ((int)((isl.list >> (4 * (k))) & 15u))
#define GEN_FOR_BLOCKS(b) for (int k_##b = 0, b = ISL_BLK(0); k_##b < isl.n; k_##b++, b = ISL_BLK(k_##b))
// (ci, gi): compact and global index of every dof of the island
#define GEN_FOR_DOFS(ci, gi) GEN_FOR_BLOCKS(b_) for (int ci = 6 * k_b_, gi = 6 * b_, e_ = gi + (b_ < gc.nb ? 6 : NDOF); gi < e_; ci++, gi++)
#define GEN_FOR_CONTACTS(ci) GEN_FOR_BLOCKS(c_) if (c_ < gc.nb) \
    for (int q_ = 0, n_ = (int)GLS(GL_INFO + c_) & 31, ci = c_ * GEN_SEG; q_ < n_; q_++, ci++)
D3IL_HD int isl_rank(const Isl isl, int b) {   // position of block b in the island's list
  int r = 0;
  for (int k = 0; k < isl.n; k++) if (ISL_BLK(k) == b) r = k;
  return r;
}

// rows of a contact over the solver dofs; the contact normal points from geom 1 to geom 2 and the row is J(body 2) - J(body 1).
// Block 1: the first cube (6 dofs, global offset o1, compact c1); block 2: the second cube or the arm (n2 = 0, 6 or 7 dofs)
struct GRow { int o1, o2, n1, n2, c1, c2; double v1[6], v2[7]; };      // n1 = 0: no first block (rod <-> static box)
// rec: the first 16 fields of the contact's record (pos[3] frame[9] dist kind a b), fetched by the caller in one batch
// RS (here and below): the build of the engine that knows contacts of the arm block (rod <-> static boxes, Inserting); with RS = false the code is the
// Sorting engine as it was - no empty-first-block guards in the row loops, no arm contact count anywhere
template <bool RS>
D3IL_HD void gen_rows(const GenConsts& gc_, const PushScratch sc, const Isl isl, const double* rec, GRow* rows) {
  D3IL_GEN_CONSTS(gc_, gc);
  const int arm0 = 6 * gc.nb;
  const int kind = (int)rec[13], a = (int)rec[14], b = (int)rec[15];
  if (RS && kind == GK_RODST) {          // static box (geom 1, fixed) -> rod (geom 2): the row is + J(arm)
#pragma unroll
    for (int rr = 0; rr < 3; rr++) {
      GRow& s = rows[rr];
#pragma unroll
      for (int k = 0; k < 6; k++) s.v1[k] = 0;
#pragma unroll
      for (int k = 0; k < 7; k++) s.v2[k] = GLS(GL_JA + 21 * (GEN_MAXNB + b) + 7 * rr + k);
      s.o1 = arm0; s.c1 = isl.m - NDOF; s.n1 = 0; s.o2 = arm0; s.n2 = 7; s.c2 = isl.m - NDOF;
    }
    return;
  }
  const int cb1 = kind == GK_STATIC ? b : a;        // first cube of the row (static: b = cube, a = static index)
  double R[9], r[3];
#pragma unroll
  for (int k = 0; k < 9; k++) R[k] = GLS(GL_R + 9 * cb1 + k);
#pragma unroll
  for (int k = 0; k < 3; k++) r[k] = rec[k] - GLS(GL_POS + 3 * cb1 + k);
  const int c1 = 6 * isl_rank(isl, cb1);
  int o2 = 0, n2 = 0, c2 = 0;
  double sign1 = -1.0;
  if (kind == GK_STATIC) sign1 = gc.st_first[a] ? 1.0 : -1.0;      // cube is geom 2 / geom 1
  else if (kind == GK_BOXBOX) { o2 = 6 * b; n2 = 6; c2 = 6 * isl_rank(isl, b); }
  else { o2 = arm0; n2 = 7; c2 = isl.m - NDOF; }
#pragma unroll
  for (int rr = 0; rr < 3; rr++) {
    GRow& s = rows[rr];
    box_row_r(R, r, rec + 3 + 3 * rr, s.v1);
#pragma unroll
    for (int k = 0; k < 6; k++) s.v1[k] *= sign1;
#pragma unroll
    for (int k = 0; k < 7; k++) s.v2[k] = 0;
    s.o1 = 6 * cb1; s.c1 = c1; s.n1 = 6; s.o2 = o2; s.n2 = n2; s.c2 = c2;
  }
  if (kind == GK_BOXBOX) {
    double R2[9], r2[3];
#pragma unroll
    for (int k = 0; k < 9; k++) R2[k] = GLS(GL_R + 9 * b + k);
#pragma unroll
    for (int k = 0; k < 3; k++) r2[k] = rec[k] - GLS(GL_POS + 3 * b + k);
#pragma unroll
    for (int rr = 0; rr < 3; rr++) {
      double t[6];
      box_row_r(R2, r2, rec + 3 + 3 * rr, t);
#pragma unroll
      for (int k = 0; k < 6; k++) rows[rr].v2[k] = t[k];
    }
  } else if (kind == GK_ROD) {
#pragma unroll
    for (int rr = 0; rr < 3; rr++)
#pragma unroll
      for (int k = 0; k < 7; k++) rows[rr].v2[k] = GLS(GL_JA + 21 * a + 7 * rr + k);
  }
}
// row . vector of the t area: globally indexed (x, velocities) or the island's compact vector at vec
template <bool RS>
D3IL_HD double grow_dot_g(const PushScratch sc, const GRow& s, int vec) {
  double acc = 0;
#pragma unroll
  for (int k = 0; k < 6; k++) if (!RS || k < s.n1) acc += s.v1[k] * GLS(vec + s.o1 + k);
#pragma unroll
  for (int k = 0; k < 7; k++) if (k < s.n2) acc += s.v2[k] * GLS(vec + s.o2 + k);
  return acc;
}
template <bool RS>
D3IL_HD double grow_dot_c(const PushScratch sc, const GRow& s, int vec) {
  double acc = 0;
#pragma unroll
  for (int k = 0; k < 6; k++) if (!RS || k < s.n1) acc += s.v1[k] * GLS(vec + s.c1 + k);
#pragma unroll
  for (int k = 0; k < 7; k++) if (k < s.n2) acc += s.v2[k] * GLS(vec + s.c2 + k);
  return acc;
}
// sum_k M(a, k) v(k) over the arm dofs; v at t-area offset va (minus the vector at vb if vb >= 0)
D3IL_HD double gen_arm_Mv(const PushScratch sc, int a, int va, int vb) {
  double s = 0;
  for (int k = 0; k < NDOF; k++) {
    double m = GLS(GL_M + (a >= k ? tri(a, k) : tri(k, a)));
    s += m * (vb >= 0 ? GLS(va + k) - GLS(vb + k) : GLS(va + k));
  }
  return s;
}
// Dense Cholesky of the island's compact Hessian (packed lower at hb, order m - always a multiple of 3), right-looking over
// 3 x 3 blocks: a block update fetches its 27 operands in one batch of LDS reads and does 27 multiply-adds in registers, so
// the LDS latency is paid once per block instead of once per element.  The nl lanes of the group share the work: every lane
// factors the diagonal block (in registers; lane 0 stores it), block rows of the panel and of the trailing update are dealt
// out round robin (a block row is only ever touched by its own lane between two syncs).
D3IL_HD bool gen_chol(const PushScratch sc, int hb, int m, int l, int nl) {
  bool ok = true;
  const int nbk = m / 3;
  for (int jb = 0; jb < nbk; jb++) {
    const int j0 = 3 * jb;
    double d00 = GLS(hb + tri(j0, j0)), d10 = GLS(hb + tri(j0 + 1, j0)), d11 = GLS(hb + tri(j0 + 1, j0 + 1));
    double d20 = GLS(hb + tri(j0 + 2, j0)), d21 = GLS(hb + tri(j0 + 2, j0 + 1)), d22 = GLS(hb + tri(j0 + 2, j0 + 2));
    if (!(d00 > 0)) { ok = false; d00 = 1; }
    const double i00 = rsqrtd(d00), l00 = d00 * i00;
    const double l10 = d10 * i00, l20 = d20 * i00;
    double t11 = d11 - l10 * l10;
    if (!(t11 > 0)) { ok = false; t11 = 1; }
    const double i11 = rsqrtd(t11), l11 = t11 * i11;
    const double l21 = (d21 - l20 * l10) * i11;
    double t22 = d22 - l20 * l20 - l21 * l21;
    if (!(t22 > 0)) { ok = false; t22 = 1; }
    const double i22 = rsqrtd(t22), l22 = t22 * i22;
    gen_sync();        // every lane has read the diagonal block
    if (l == 0) {
      GLS(hb + tri(j0, j0)) = l00; GLS(hb + tri(j0 + 1, j0)) = l10; GLS(hb + tri(j0 + 1, j0 + 1)) = l11;
      GLS(hb + tri(j0 + 2, j0)) = l20; GLS(hb + tri(j0 + 2, j0 + 1)) = l21; GLS(hb + tri(j0 + 2, j0 + 2)) = l22;
    }
    // panel: L(ib, jb) = H(ib, jb) L(jb, jb)^-T
    for (int ib = jb + 1 + l; ib < nbk; ib += nl) {
      double a[3][3];
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) a[r][c] = GLS(hb + tri(3 * ib + r, j0 + c));
#pragma unroll
      for (int r = 0; r < 3; r++) {
        const double x0 = a[r][0] * i00;
        const double x1 = (a[r][1] - x0 * l10) * i11;
        const double x2 = (a[r][2] - x0 * l20 - x1 * l21) * i22;
        GLS(hb + tri(3 * ib + r, j0)) = x0; GLS(hb + tri(3 * ib + r, j0 + 1)) = x1; GLS(hb + tri(3 * ib + r, j0 + 2)) = x2;
      }
    }
    gen_sync();        // panel complete
    // trailing update: H(ib, kb) -= L(ib, jb) L(kb, jb)^T for jb < kb <= ib
    for (int ib = jb + 1 + l; ib < nbk; ib += nl) {
      double li[3][3];
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) li[r][c] = GLS(hb + tri(3 * ib + r, j0 + c));
      for (int kb = jb + 1; kb <= ib; kb++) {
        double lk[3][3], h[3][3];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
          for (int c = 0; c < 3; c++) {
            lk[r][c] = GLS(hb + tri(3 * kb + r, j0 + c));
            h[r][c] = (kb < ib || c <= r) ? GLS(hb + tri(3 * ib + r, 3 * kb + c)) : 0.0;
          }
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
          for (int c = 0; c < 3; c++) h[r][c] -= li[r][0] * lk[c][0] + li[r][1] * lk[c][1] + li[r][2] * lk[c][2];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
          for (int c = 0; c < 3; c++) if (kb < ib || c <= r) GLS(hb + tri(3 * ib + r, 3 * kb + c)) = h[r][c];
      }
    }
    gen_sync();        // trailing update complete
  }
  return ok;
}
// the two triangular solves on the compact vector at vec; executed by every lane of the group with identical operands
// (identical stores), so that no hand-over is needed
D3IL_HD void gen_chol_solve(const PushScratch sc, int hb, int m, int vec) {
  const int nbk = m / 3;
  for (int ib = 0; ib < nbk; ib++) {          // L y = b, block row by block row; the sums accumulate in registers
    const int i0 = 3 * ib;
    double y0 = GLS(vec + i0), y1 = GLS(vec + i0 + 1), y2 = GLS(vec + i0 + 2);
#pragma unroll 2
    for (int kb = 0; kb < ib; kb++) {
      const double v0 = GLS(vec + 3 * kb), v1 = GLS(vec + 3 * kb + 1), v2 = GLS(vec + 3 * kb + 2);
      const int r0 = hb + tri(i0, 3 * kb), r1 = hb + tri(i0 + 1, 3 * kb), r2 = hb + tri(i0 + 2, 3 * kb);
      y0 -= GLS(r0) * v0 + GLS(r0 + 1) * v1 + GLS(r0 + 2) * v2;
      y1 -= GLS(r1) * v0 + GLS(r1 + 1) * v1 + GLS(r1 + 2) * v2;
      y2 -= GLS(r2) * v0 + GLS(r2 + 1) * v1 + GLS(r2 + 2) * v2;
    }
    const double l00 = GLS(hb + tri(i0, i0)), l10 = GLS(hb + tri(i0 + 1, i0)), l11 = GLS(hb + tri(i0 + 1, i0 + 1));
    const double l20 = GLS(hb + tri(i0 + 2, i0)), l21 = GLS(hb + tri(i0 + 2, i0 + 1)), l22 = GLS(hb + tri(i0 + 2, i0 + 2));
    y0 = y0 * rcpd(l00); y1 = (y1 - l10 * y0) * rcpd(l11); y2 = (y2 - l20 * y0 - l21 * y1) * rcpd(l22);
    GLS(vec + i0) = y0; GLS(vec + i0 + 1) = y1; GLS(vec + i0 + 2) = y2;
  }
  for (int ib = nbk - 1; ib >= 0; ib--) {      // L' x = y
    const int i0 = 3 * ib;
    double x0 = GLS(vec + i0), x1 = GLS(vec + i0 + 1), x2 = GLS(vec + i0 + 2);
#pragma unroll 2
    for (int kb = ib + 1; kb < nbk; kb++) {
      const double v0 = GLS(vec + 3 * kb), v1 = GLS(vec + 3 * kb + 1), v2 = GLS(vec + 3 * kb + 2);
      const int r0 = hb + tri(3 * kb, i0), r1 = hb + tri(3 * kb + 1, i0), r2 = hb + tri(3 * kb + 2, i0);
      x0 -= GLS(r0) * v0 + GLS(r1) * v1 + GLS(r2) * v2;
      x1 -= GLS(r0 + 1) * v0 + GLS(r1 + 1) * v1 + GLS(r2 + 1) * v2;
      x2 -= GLS(r0 + 2) * v0 + GLS(r1 + 2) * v1 + GLS(r2 + 2) * v2;
    }
    const double l00 = GLS(hb + tri(i0, i0)), l10 = GLS(hb + tri(i0 + 1, i0)), l11 = GLS(hb + tri(i0 + 1, i0 + 1));
    const double l20 = GLS(hb + tri(i0 + 2, i0)), l21 = GLS(hb + tri(i0 + 2, i0 + 1)), l22 = GLS(hb + tri(i0 + 2, i0 + 2));
    x2 = x2 * rcpd(l22); x1 = (x1 - l21 * x2) * rcpd(l11); x0 = (x0 - l10 * x1 - l20 * x2) * rcpd(l00);
    GLS(vec + i0) = x0; GLS(vec + i0 + 1) = x1; GLS(vec + i0 + 2) = x2;
  }
}


// accumulation into the t area by several lanes of a group at once (LDS atomic add; the conflicting lanes of one instruction
// are served in lane order, so the sum is reproducible); the host build runs the lanes one after the other
#if defined(__HIP_DEVICE_COMPILE__)
#define GLS_ADD(i, v) ((void)__hip_atomic_fetch_add(&GLS(i), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT))
#else
#define GLS_ADD(i, v) (GLS(i) += (v))
#endif

// Newton solve of one island by the nl lanes of the group together (lane l); x in / out at GL_X (island dofs).  cpk: contact
// counts of the island's blocks, 5 bits each in list order.  Work is dealt out by contact (lane l takes contacts l, l + nl, ..
// of the island's concatenated segments) and by block row (Cholesky); everything else is computed by every lane from the same
// LDS data, so all lanes take the same decisions.  Returns false when it did not converge.
template <bool RS>
D3IL_NOINLINE inline bool gen_solve(const GenConsts& gc_, const PushScratch sc, const Isl isl, unsigned cpk, int l, int nl) {
  D3IL_GEN_CONSTS(gc_, gc);
  const int arm0 = 6 * gc.nb, ca = isl.m - NDOF;       // global / compact offset of the arm block (when isl.arm)
  const int hb = GL_H + isl.hoff, vg = GL_G + isl.voff, vp = GL_P + isl.voff;
  const double impr = gc.impratio, mu_scale = sqrt(1 / fmax(1e-15, impr));
  int ntot = 0;
  for (int k = 0; k < isl.n; k++) ntot += (int)((cpk >> (5 * k)) & 31u);
  auto locate = [&](int t) {      // record index of the island's t-th contact
    int ci = 0;
    bool found = false;
    for (int k = 0; k < isl.n; k++) {
      const int n = (int)((cpk >> (5 * k)) & 31u);
      if (!found && t < n) { const int blk = ISL_BLK(k); ci = (!RS || blk < gc.nb ? blk : GEN_ARMSEG) * GEN_SEG + t; found = true; }      // the arm block's contacts: rod <-> static boxes
      t -= n;
    }
    return ci;
  };
  auto dof_of = [&](int ci) {     // global dof of compact dof ci
    return (isl.arm && ci >= ca) ? arm0 + ci - ca : 6 * ISL_BLK(ci / 6) + ci % 6;
  };
  PUSH_TIC;
  for (int t = l; t < ntot; t += nl) {   // reference acceleration and regularisation
    const int base = GG_CON + locate(t) * GREC;
    double rec[16];
#pragma unroll
    for (int k = 0; k < 16; k++) rec[k] = GRS(base + k);
    const int kind = (int)rec[13], set = kind == GK_STATIC ? (int)rec[14] : (kind == GK_BOXBOX ? gc.set_bb : (!RS || kind == GK_ROD ? gc.set_rod : gc.ns + 2 + (int)rec[14]));
    const double dist = rec[12];
    double imp = impedance(gc.ct_solimp[set], dist);
    double invw = kind == GK_STATIC ? gc.box_invw_t : (kind == GK_BOXBOX ? 2 * gc.box_invw_t : (!RS || kind == GK_ROD ? gc.box_invw_t + gc.rod_invw : gc.rod_invw));
    GRow rows[3];
    gen_rows<RS>(gc, sc, isl, rec, rows);
    double v0 = grow_dot_g<RS>(sc, rows[0], GL_VEL), v1 = grow_dot_g<RS>(sc, rows[1], GL_VEL), v2 = grow_dot_g<RS>(sc, rows[2], GL_VEL);
    GRS(base + 16) = -gc.ct_B[set] * v0 - gc.ct_K[set] * imp * dist;
    GRS(base + 17) = -gc.ct_B[set] * v1; GRS(base + 18) = -gc.ct_B[set] * v2;
    GRS(base + 19) = 1 / fmax(1e-15, (1 - imp) / imp * invw);
    GRS(base + 20) = gc.ct_fric[set];
  }
  gen_sync();
  bool converged = false;
  int buf = 0;
  PUSH_TOC(3);
  D3IL_STAT(g_stats.newton_calls++);
  PUSH_CNT(7);
  D3IL_STAT(g_stats.eig_calls += isl.m);
  for (int it = 0; it < 60 && !converged; it++) {
    D3IL_STAT(g_stats.newton_iters++);
    D3IL_STAT(g_stats.ik_calls += isl.m * isl.m * isl.m / 6);
    // gradient (compact, at vg) and Hessian (compact, at hb) at x: smooth part entry by entry, then the limit rows (lane 0),
    // then the contacts (accumulated by all lanes)
    for (int i = l, nh = isl.m * (isl.m + 1) / 2; i < nh; i += nl) GLS(hb + i) = 0;
    gen_sync();
    for (int ci = l; ci < isl.m; ci += nl) {
      const int gi = dof_of(ci);
      if (gi < arm0) { const double mm = (gi % 6) < 3 ? gc.box_mass : gc.box_inertia; GLS(vg + ci) = mm * (GLS(GL_X + gi) - GLS(GL_A0 + gi)); GLS(hb + tri(ci, ci)) = mm; }
      else {
        GLS(vg + ci) = gen_arm_Mv(sc, gi - arm0, GL_X + arm0, GL_A0 + arm0);
        for (int k = 0; k <= gi - arm0; k++) GLS(hb + tri(ci, ca + k)) = GLS(GL_M + tri(gi - arm0, k));
      }
    }
    gen_sync();
    if (isl.arm && l == 0)
      for (int k = 0; k < NDOF; k++) {
        double sign = GLS(GL_LIM + 3 * k), D = GLS(GL_LIM + 3 * k + 1), aref = GLS(GL_LIM + 3 * k + 2);
        if (sign != 0) {
          double jar = sign * GLS(GL_X + arm0 + k) - aref;
          if (jar < 0) { GLS(vg + ca + k) += sign * D * jar; GLS(hb + tri(ca + k, ca + k)) += D; }
        }
      }
    gen_sync();
    for (int t = l; t < ntot; t += nl) {
      const int base = GG_CON + locate(t) * GREC;
      double rec[21];
#pragma unroll
      for (int k = 0; k < 21; k++) rec[k] = GRS(base + k);
      GRow rows[3];
      gen_rows<RS>(gc, sc, isl, rec, rows);
      double jar[3], force[3], Hc[9];
#pragma unroll
      for (int r = 0; r < 3; r++) { jar[r] = grow_dot_g<RS>(sc, rows[r], GL_X) - rec[16 + r]; GRS(base + 22 + r) = jar[r]; }
      const double Dn = rec[19], fric = rec[20];
      cone_eval(jar, Dn, Dn * impr, fric * mu_scale, fric, force, Hc);
      if (force[0] == 0 && force[1] == 0 && force[2] == 0) continue;
      const int c1 = rows[0].c1, c2 = rows[0].c2, n1 = RS ? rows[0].n1 : 6, n2 = rows[0].n2;
#pragma unroll
      for (int k = 0; k < 6; k++) if (!RS || k < n1) GLS_ADD(vg + c1 + k, -(rows[0].v1[k] * force[0] + rows[1].v1[k] * force[1] + rows[2].v1[k] * force[2]));
#pragma unroll
      for (int k = 0; k < 7; k++) if (k < n2) GLS_ADD(vg + c2 + k, -(rows[0].v2[k] * force[0] + rows[1].v2[k] * force[1] + rows[2].v2[k] * force[2]));
      // H += J' Hc J, block (1,1), then (2,1) and (2,2); c2 > c1 for every contact kind (second cube after the first, arm last)
#pragma unroll
      for (int a = 0; a < 6; a++) if (!RS || a < n1) {
        double ta[3];
#pragma unroll
        for (int r = 0; r < 3; r++) ta[r] = Hc[3 * r] * rows[0].v1[a] + Hc[3 * r + 1] * rows[1].v1[a] + Hc[3 * r + 2] * rows[2].v1[a];
#pragma unroll
        for (int b = 0; b <= a; b++) GLS_ADD(hb + tri(c1 + a, c1 + b), ta[0] * rows[0].v1[b] + ta[1] * rows[1].v1[b] + ta[2] * rows[2].v1[b]);
      }
#pragma unroll
      for (int a = 0; a < 7; a++) if (a < n2) {
        double ta[3];
#pragma unroll
        for (int r = 0; r < 3; r++) ta[r] = Hc[3 * r] * rows[0].v2[a] + Hc[3 * r + 1] * rows[1].v2[a] + Hc[3 * r + 2] * rows[2].v2[a];
#pragma unroll
        for (int b = 0; b < 6; b++) if (!RS || b < n1) GLS_ADD(hb + tri(c2 + a, c1 + b), ta[0] * rows[0].v1[b] + ta[1] * rows[1].v1[b] + ta[2] * rows[2].v1[b]);
#pragma unroll
        for (int b = 0; b < 7; b++) if (b <= a) GLS_ADD(hb + tri(c2 + a, c2 + b), ta[0] * rows[0].v2[b] + ta[1] * rows[1].v2[b] + ta[2] * rows[2].v2[b]);
      }
    }
    gen_sync();
    {
      double gm = 0;
      for (int k = 0; k < isl.m; k++) gm = fmax(gm, fabs(GLS(vg + k)));
      if (gm <= PUSH_GRAD_TOL) { converged = true; break; }
    }
    PUSH_TOC(4);
    if (!gen_chol(sc, hb, isl.m, l, nl)) return false;
    for (int k = 0; k < isl.m; k++) GLS(vp + k) = -GLS(vg + k);
    gen_chol_solve(sc, hb, isl.m, vp);
    gen_sync();
    PUSH_TOC(5);
    double pMp = 0, pMa = 0, gTp = 0;
    for (int ci = 0; ci < isl.m; ci++) {
      const int gi = dof_of(ci);
      double s, sa, pi = GLS(vp + ci);
      if (gi < arm0) { const double mm = (gi % 6) < 3 ? gc.box_mass : gc.box_inertia; s = mm * pi; sa = mm * (GLS(GL_X + gi) - GLS(GL_A0 + gi)); }
      else { s = gen_arm_Mv(sc, gi - arm0, vp + ca, -1); sa = gen_arm_Mv(sc, gi - arm0, GL_X + arm0, GL_A0 + arm0); }
      pMp += pi * s; pMa += pi * sa; gTp += GLS(vg + ci) * pi;
    }
    for (int t = l; t < ntot; t += nl) {
      const int base = GG_CON + locate(t) * GREC;
      double rec[16];
#pragma unroll
      for (int k = 0; k < 16; k++) rec[k] = GRS(base + k);
      GRow rows[3];
      gen_rows<RS>(gc, sc, isl, rec, rows);
#pragma unroll
      for (int r = 0; r < 3; r++) GRS(base + 25 + r) = grow_dot_c<RS>(sc, rows[r], vp);
    }
    gen_sync();
    PUSH_TOC(6);
    double alpha = 1, lo = 0, hi = -1, best = 1, wprev = 1e300;
    for (int ls = 0; ls < 50; ls++) {
      D3IL_STAT(g_stats.ls_iters++);
      double p1 = 0, p2 = 0;      // this lane's share of the contact terms of phi'(alpha), phi''(alpha)
      for (int t = l; t < ntot; t += nl) {
        const int base = GG_CON + locate(t) * GREC;
        double rec[9];
#pragma unroll
        for (int k = 0; k < 9; k++) rec[k] = GRS(base + 19 + k);     // Dn fric set jar[3] jp[3]
        double jp[3] = {rec[6], rec[7], rec[8]};
        double jt[3] = {rec[3] + alpha * jp[0], rec[4] + alpha * jp[1], rec[5] + alpha * jp[2]}, ft[3], Hc[9];
        const double Dn = rec[0], fric = rec[1];
        cone_eval(jt, Dn, Dn * impr, fric * mu_scale, fric, ft, Hc);
#pragma unroll
        for (int r = 0; r < 3; r++) { p1 -= ft[r] * jp[r];
#pragma unroll
          for (int q = 0; q < 3; q++) p2 += jp[r] * Hc[3 * r + q] * jp[q]; }
      }
      double d1 = pMa + alpha * pMp, d2 = pMp;
#if defined(__HIP_DEVICE_COMPILE__)
      if (nl == GEN_WAVE) {      // the whole wave on this island (step kernel, gen_kernels.h): butterfly sums - the same value in every lane (the additions commute), no LDS slots
        gen_sync();
#pragma unroll
        for (int o = 1; o < GEN_WAVE; o <<= 1) { p1 += __shfl_xor(p1, o); p2 += __shfl_xor(p2, o); }
        gen_sync();
        d1 += p1; d2 += p2;
      } else
#endif
      {
        GLS(GL_RED + GL_NRED * buf + 2 * l) = p1; GLS(GL_RED + GL_NRED * buf + 2 * l + 1) = p2;
        gen_sync();
        for (int j = 0; j < nl; j++) { d1 += GLS(GL_RED + GL_NRED * buf + 2 * j); d2 += GLS(GL_RED + GL_NRED * buf + 2 * j + 1); }
        buf ^= 1;
      }
      if (isl.arm)
        for (int k = 0; k < NDOF; k++) {
          double sign = GLS(GL_LIM + 3 * k), D = GLS(GL_LIM + 3 * k + 1), aref = GLS(GL_LIM + 3 * k + 2);
          if (sign != 0) {
            double jp = sign * GLS(vp + ca + k), jar = sign * GLS(GL_X + arm0 + k) - aref + alpha * jp;
            if (jar < 0) { d1 += D * jar * jp; d2 += D * jp * jp; }
          }
        }
      best = alpha;
      if (ls == 0 && d1 <= D3IL_TOL.ls_full * fabs(gTp)) break;
      if (fabs(d1) <= D3IL_TOL.ls_c2 * fabs(gTp) || fabs(d1) <= D3IL_TOL.ls_rel * d2 * alpha || fabs(d1) < 1e-14 * fmax(1.0, fabs(pMa))) break;
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
    PUSH_TOC(7);
    double smax = 0, xmax = 0;
    for (int ci = 0; ci < isl.m; ci++) {
      const int gi = dof_of(ci);
      const double dxk = best * GLS(vp + ci), xo = GLS(GL_X + gi);
      smax = fmax(smax, fabs(dxk)); xmax = fmax(xmax, fabs(xo + dxk));
    }
    gen_sync();        // every lane has read the old iterate
    for (int ci = l; ci < isl.m; ci += nl) { const int gi = dof_of(ci); GLS(GL_X + gi) += best * GLS(vp + ci); }
    gen_sync();
    if (smax <= 1e-12 * (1 + xmax) || (best == 1.0 && smax <= D3IL_TOL.step_rel * (1 + xmax))) converged = true;
  }
  return converged;
}
}  // namespace d3il
#include "gen_tree.h"
namespace d3il {

// ---- phase 1 (lane 0): arm forward pass.  Publishes M, qacc_smooth, velocities, limit rows, rod pose; leaves the arm-alone
// solution (finger limit rows by the exact active-set solution, as in panda_step.h) at GL_X
template <class C>
D3IL_HD void gen_phase1(const C& c0, const GenConsts& gc_, EnvState& st, const PushScratch sc, const double* tau, const double* ffing) {
  D3IL_GEN_CONSTS(gc_, gc);
  D3IL_REFRESH(c0, c);
  const int arm0 = 6 * gc.nb;
  DynOut dyn;
  dynamics(c0, st.q, st.v, dyn);
  double fs[NDOF];
#pragma unroll
  for (int k = 0; k < NARM; k++) fs[k] = clampd(tau[k] + st.bias[k], c.force_lo[k], c.force_hi[k]) - dyn.bias[k];
#pragma unroll
  for (int k = 0; k < NFING; k++) fs[NARM + k] = clampd(ffing[k], c.force_lo[NARM + k], c.force_hi[NARM + k]) - dyn.bias[NARM + k] - c.f_damping[k] * st.v[NARM + k];
#pragma unroll
  for (int k = 0; k < NARM; k++) st.bias[k] = dyn.bias[k];
  {
    double t[3], rodc[3], rodu[3];
    mulE(dyn.R7, c.tcp7, t);
    st.tcp[0] = dyn.p7[0] + t[0]; st.tcp[1] = dyn.p7[1] + t[1]; st.tcp[2] = dyn.p7[2] + t[2];
    mulE(dyn.R7, c.rod_c7, rodc);
    mulE(dyn.R7, c.rod_u7, rodu);
#pragma unroll
    for (int k = 0; k < 3; k++) { GLS(GL_ROD + k) = rodc[k] + dyn.p7[k]; GLS(GL_ROD + 3 + k) = rodu[k]; }
  }
  double L[45], d[NDOF], id[NDOF], a0[NDOF];
  if (!ldl9(dyn.M, L, d, id)) st.flags |= F_SOLVER_FAIL;
#pragma unroll
  for (int k = 0; k < NDOF; k++) a0[k] = fs[k];
  ldl9_solve(L, id, a0);
#pragma unroll
  for (int i = 0; i < 45; i++) GLS(GL_M + i) = dyn.M[i];
#pragma unroll
  for (int k = 0; k < NDOF; k++) { GLS(GL_VEL + arm0 + k) = st.v[k]; GLS(GL_A0 + arm0 + k) = a0[k]; }
  bool arm_lim = false;
  double fc[NDOF];
#pragma unroll
  for (int k = 0; k < NDOF; k++) {
    fc[k] = 0;
    double dlo = st.q[k] - c.jnt_range[k][0], dhi = c.jnt_range[k][1] - st.q[k];
    double sign = 0, dist = 0, D = 0, aref = 0;
    if (dlo < c.lim_margin[k]) { sign = 1; dist = dlo; }
    else if (dhi < c.lim_margin[k]) { sign = -1; dist = dhi; }
    if (sign != 0) {
      double imp = impedance(c.lim_solimp[k], dist - c.lim_margin[k]);
      D = 1 / fmax(1e-15, (1 - imp) / imp * c.dof_invweight0[k]);
      aref = -c.lim_B[k] * (sign * st.v[k]) - c.lim_K[k] * imp * (dist - c.lim_margin[k]);
      if (k < NARM) arm_lim = true;
    }
    GLS(GL_LIM + 3 * k) = sign; GLS(GL_LIM + 3 * k + 1) = D; GLS(GL_LIM + 3 * k + 2) = aref;
  }
  GLS(GL_INFO + 4) = arm_lim ? 1.0 : 0.0;
  {   // finger rows alone: 2 x 2 active-set solution in force space
    double s0 = GLS(GL_LIM + 3 * 7), s1 = GLS(GL_LIM + 3 * 8);
    if (s0 != 0 || s1 != 0) {
      double D0 = GLS(GL_LIM + 3 * 7 + 1), D1 = GLS(GL_LIM + 3 * 8 + 1), ar0 = GLS(GL_LIM + 3 * 7 + 2), ar1 = GLS(GL_LIM + 3 * 8 + 2);
      double l87 = L[tri(8, 7)];
      double W00 = id[7] + l87 * l87 * id[8], W01 = -l87 * id[8], W11 = id[8];
      double r0 = s0 * a0[7] - ar0, r1 = s1 * a0[8] - ar1;
      double G00 = W00 * s0 * s0, G01 = W01 * s0 * s1, G11 = W11 * s1 * s1;
      double f0 = 0, f1 = 0;
      bool have0 = s0 != 0, have1 = s1 != 0, done = false;
      if (have0 && have1) {
        double a = 1 + D0 * G00, b = D0 * G01, cc = D1 * G01, dd = 1 + D1 * G11;
        double det = a * dd - b * cc, y0 = -D0 * r0, y1 = -D1 * r1;
        double g0 = (dd * y0 - b * y1) / det, g1 = (a * y1 - cc * y0) / det;
        if (g0 > 0 && g1 > 0) { f0 = g0; f1 = g1; done = true; }
      }
      if (!done && have0) {
        double g0 = -D0 * r0 / (1 + D0 * G00);
        if (g0 > 0 && (!have1 || r1 + G01 * g0 >= 0)) { f0 = g0; f1 = 0; done = true; }
      }
      if (!done && have1) {
        double g1 = -D1 * r1 / (1 + D1 * G11);
        if (g1 > 0 && (!have0 || r0 + G01 * g1 >= 0)) { f1 = g1; f0 = 0; done = true; }
      }
      fc[7] = s0 * f0; fc[8] = s1 * f1;
    }
  }
  double xa[NDOF];
#pragma unroll
  for (int k = 0; k < NDOF; k++) xa[k] = fs[k] + fc[k];
  ldl9_solve(L, id, xa);
#pragma unroll
  for (int k = 0; k < NDOF; k++) GLS(GL_X + arm0 + k) = xa[k];
}

// sub: -1 = this lane stores every field; 0 / 1 = the lane is one of the two sub-lanes of its cube's pair, both run the collision phase with identical
// data and each stores the fields of its parity - one store instruction of the wave then covers field pairs of eight environments, i.e. whole lines
D3IL_HD void gen_put(const GenConsts& gc_, const PushScratch sc, int cube, int& cnt, unsigned& fl, const double* rec, int kind, int a, int b, int set, int sub = -1) {
  D3IL_GEN_CONSTS(gc_, gc);
  if (cnt >= GEN_SEG) { fl |= PF_CON_OVERFLOW; return; }
  const int base = GG_CON + (cube * GEN_SEG + cnt) * GREC;
  double n[3] = {rec[4], rec[5], rec[6]}, t1[3], t2[3];
  make_frame(n, t1, t2);
  const double f[22] = {rec[1], rec[2], rec[3], n[0], n[1], n[2], t1[0], t1[1], t1[2], t2[0], t2[1], t2[2], rec[0], (double)kind, (double)a, (double)b, 0, 0, 0, 0,
                        gc.ct_fric[set], kind == GK_STATIC && gc.st_first[a] ? 1.0 : -1.0};      // friction and sign with the record: no table look-ups by record content in the solver's loops
  if (sub < 0) {
#pragma unroll
    for (int k = 0; k < 22; k++) if (k < 16 || k >= 20) GRS(base + k) = f[k];
  } else {
#pragma unroll
    for (int k = 0; k < 22; k += 2) if (k < 16 || k >= 20) GRS(base + k + sub) = sub ? f[k + 1] : f[k];
  }
  cnt++;
}
// ---- phase 2 (lane c): cube c's pose into the t area, collision against the static boxes.  Returns the contact count.
D3IL_NOINLINE inline int gen_phase2(const GenConsts& gc_, const PushScratch sc, int c, const double* gravity, unsigned& fl, int sub = -1) {
  D3IL_GEN_CONSTS(gc_, gc);
  double pc[3], Rc[9];
  {
    double q[4] = {GBX(c, 3), GBX(c, 4), GBX(c, 5), GBX(c, 6)};
    quat2mat(q, Rc);
#pragma unroll
    for (int k = 0; k < 9; k++) GLS(GL_R + 9 * c + k) = Rc[k];
#pragma unroll
    for (int k = 0; k < 3; k++) { pc[k] = GBX(c, k); GLS(GL_POS + 3 * c + k) = pc[k]; }
#pragma unroll
    for (int k = 0; k < 6; k++) { GLS(GL_VEL + 6 * c + k) = GBX(c, 7 + k); GLS(GL_A0 + 6 * c + k) = k < 3 ? gravity[k] : 0.0; }
    if (pc[0] < gc.ws_lo[0] || pc[0] > gc.ws_hi[0] || pc[1] < gc.ws_lo[1] || pc[1] > gc.ws_hi[1]) fl |= PF_OFF_TABLE;
  }
  const double rcirc2 = gc.box_half[0] * gc.box_half[0] + gc.box_half[1] * gc.box_half[1] + gc.box_half[2] * gc.box_half[2];
  int cnt = 0;
  // (contacts go from the collider straight into the records - box_box_emit's callback; an array of candidates [8][7] indexed at run time lived in scratch:
  // 112 + 112 scratch instructions per wave and sub-step whenever a lane had a contact, a quarter of the resting regime's memory traffic in round 5)
  // the frame beams (lab_surrounding.xml:3-114: a 19 mm rim around the table top) come last in the list and are skipped for a cube well inside the table
  const bool near_edge = pc[0] < gc.in_lo[0] || pc[0] > gc.in_hi[0] || pc[1] < gc.in_lo[1] || pc[1] > gc.in_hi[1];
  const int ns = near_edge ? gc.ns : gc.ns_core;
  for (int s = 0; s < ns; s++) {
    double d2 = 0;   // sphere against the static box (in its frame): cheap exact rejection
#pragma unroll
    for (int i = 0; i < 3; i++) {
      double x = (pc[0] - gc.st_c[s][0]) * gc.st_R[s][i] + (pc[1] - gc.st_c[s][1]) * gc.st_R[s][3 + i] + (pc[2] - gc.st_c[s][2]) * gc.st_R[s][6 + i];
      double e = fabs(x) - gc.st_h[s][i];
      if (e > 0) d2 += e * e;
    }
    if (d2 > rcirc2) continue;
    auto put = [&](double dist, const double* pos, const double* nrm) {
      const double r7[7] = {dist, pos[0], pos[1], pos[2], nrm[0], nrm[1], nrm[2]};
      gen_put(gc, sc, c, cnt, fl, r7, GK_STATIC, s, c, s, sub);
    };
    if (gc.st_first[s]) box_box_emit(gc.st_c[s], gc.st_R[s], gc.st_h[s], pc, Rc, gc.box_half, 0.0, 8, put);
    else box_box_emit(pc, Rc, gc.box_half, gc.st_c[s], gc.st_R[s], gc.st_h[s], 0.0, 8, put);
  }
  return cnt;
}
// ---- phase 3 (lane c): cube c against the cubes after it and against the rod; publishes the cube's info word
D3IL_NOINLINE inline void gen_phase3(const GenConsts& gc_, const PushScratch sc, int c, int cnt, double rod_r, double rod_h, unsigned& fl, int sub = -1) {
  D3IL_GEN_CONSTS(gc_, gc);
  const double rcirc2 = gc.box_half[0] * gc.box_half[0] + gc.box_half[1] * gc.box_half[1] + gc.box_half[2] * gc.box_half[2];
  double pc[3], Rc[9];
#pragma unroll
  for (int k = 0; k < 3; k++) pc[k] = GLS(GL_POS + 3 * c + k);
#pragma unroll
  for (int k = 0; k < 9; k++) Rc[k] = GLS(GL_R + 9 * c + k);
  unsigned partners = 0, rod = 0;
  for (int d = c + 1; d < gc.nb; d++) {
    double pd[3], Rd[9], dd = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) { pd[k] = GLS(GL_POS + 3 * d + k); dd += (pc[k] - pd[k]) * (pc[k] - pd[k]); }
    if (dd > 4 * rcirc2) continue;
#pragma unroll
    for (int k = 0; k < 9; k++) Rd[k] = GLS(GL_R + 9 * d + k);
    const int first = cnt;
    box_box_emit(pc, Rc, gc.box_half, pd, Rd, gc.box_half, 0.0, 8, [&](double dist, const double* pos, const double* nrm) {
      const double r7[7] = {dist, pos[0], pos[1], pos[2], nrm[0], nrm[1], nrm[2]};
      gen_put(gc, sc, c, cnt, fl, r7, GK_BOXBOX, c, d, gc.set_bb, sub);
    });
    if (cnt > first) { partners |= 1u << d; GLS(GL_PAIR + gt_pair(c, d)) = (double)((unsigned)first | ((unsigned)(cnt - first) << 5)); }
  }
  {
    double rodc[3], rodu[3], r1[7];
#pragma unroll
    for (int k = 0; k < 3; k++) { rodc[k] = GLS(GL_ROD + k); rodu[k] = GLS(GL_ROD + 3 + k); }
    double w[3] = {pc[0] - rodc[0], pc[1] - rodc[1], pc[2] - rodc[2]};
    double t = clampd(dot3(w, rodu), -rod_h, rod_h);
    double e[3] = {w[0] - t * rodu[0], w[1] - t * rodu[1], w[2] - t * rodu[2]};
    double rr = sqrt(rcirc2) + rod_r;
    if (dot3(e, e) < rr * rr && cyl_box(rodc, rodu, rod_r, rod_h, pc, Rc, gc.box_half, 0.0, r1)) {   // cube is geom 1: normal cube -> rod
      const int before = cnt;
      gen_put(gc, sc, c, cnt, fl, r1, GK_ROD, c, 0, gc.set_rod, sub);
      if (cnt > before) rod = 1;
    }
  }
  GLS(GL_INFO + c) = (double)((unsigned)cnt | (partners << 5) | (rod << 9));
}
// ---- phase 3r (lane l of nl, only when gc.rod_static): the rod against the static boxes l, l + nl, ..  A box is skipped when the rod's capsule
// lies beyond one of its faces (exact rejection in the box frame); what remains goes through cyl_box.  The lane parks its finds in its own
// slots [GEN_ARMLANE l, ..) of the arm segment and publishes their number; lane 0 packs them in phase 3b.
D3IL_NOINLINE inline void gen_phase3r(const GenConsts& gc_, const PushScratch sc, int l, int nl, double rod_r, double rod_h, unsigned& fl) {
  D3IL_GEN_CONSTS(gc_, gc);
  double rodc[3], rodu[3];
#pragma unroll
  for (int k = 0; k < 3; k++) { rodc[k] = GLS(GL_ROD + k); rodu[k] = GLS(GL_ROD + 3 + k); }
  int cnt = 0;
  for (int s = l; s < gc.ns_core; s += nl) {
    if (!gc.st_rod[s]) continue;
    bool apart = false;
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const double x = (rodc[0] - gc.st_c[s][0]) * gc.st_R[s][i] + (rodc[1] - gc.st_c[s][1]) * gc.st_R[s][3 + i] + (rodc[2] - gc.st_c[s][2]) * gc.st_R[s][6 + i];
      const double u = rodu[0] * gc.st_R[s][i] + rodu[1] * gc.st_R[s][3 + i] + rodu[2] * gc.st_R[s][6 + i];
      apart = apart || fabs(x) - rod_h * fabs(u) > gc.st_h[s][i] + rod_r;
    }
    if (apart) continue;
    double r1[7];
    if (!cyl_box(rodc, rodu, rod_r, rod_h, gc.st_c[s], gc.st_R[s], gc.st_h[s], 0.0, r1)) continue;      // the box is geom 1: normal box -> rod
    if (cnt >= GEN_ARMLANE) { fl |= PF_CON_OVERFLOW; continue; }
    int slot = GEN_ARMLANE * l + cnt;
    gen_put(gc, sc, GEN_ARMSEG, slot, fl, r1, GK_RODST, s, 0, gc.ns + 2 + s);
    cnt++;
  }
  GLS(GL_INFO + 9 + l) = (double)cnt;
}
// ---- phase 3b (lane 0): arm Jacobian rows of the rod contacts (the last record of a cube's segment; the packed rod <-> static contacts)
template <bool RS, class C>
D3IL_HD void gen_phase3b(const C& c0, const GenConsts& gc_, const EnvState& st, const PushScratch sc, int nl, unsigned& fl) {
  D3IL_GEN_CONSTS(gc_, gc);
  bool any = false;
  for (int b = 0; b < gc.nb; b++) any = any || (((unsigned)GLS(GL_INFO + b) >> 9) & 1);
  int narm = 0;
  if (RS && gc.rod_static) {          // pack the lanes' finds to the front of the arm segment
    for (int l = 0; l < nl; l++)
      for (int j = 0, m = (int)GLS(GL_INFO + 9 + l); j < m; j++) {
        if (narm >= GEN_ARMCON) { fl |= PF_CON_OVERFLOW; continue; }
        const int src = GG_CON + (GEN_ARMSEG * GEN_SEG + GEN_ARMLANE * l + j) * GREC, dst = GG_CON + (GEN_ARMSEG * GEN_SEG + narm) * GREC;
        if (src != dst) for (int k = 0; k < 22; k++) GRS(dst + k) = GRS(src + k);
        GRS(dst + 15) = (double)narm;
        narm++;
      }
  }
  if (RS) GLS(GL_INFO + 8) = (double)narm;
  if (!any && narm == 0) return;
  double sn[NARM], cs[NARM], R7[9], p7[3], ax[NARM][3], og[NARM][3];
#pragma unroll
  for (int i = 0; i < NARM; i++) sincos(st.q[i], &sn[i], &cs[i]);
  world_chain(c0, sn, cs, R7, p7, ax, og);
  for (int b = 0; b < gc.nb; b++) {
    unsigned info = (unsigned)GLS(GL_INFO + b);
    if (!((info >> 9) & 1)) continue;
    const int base = GG_CON + (b * GEN_SEG + (int)(info & 31) - 1) * GREC;
    double p[3] = {GRS(base), GRS(base + 1), GRS(base + 2)};
    for (int k = 0; k < NARM; k++) {
      double dd[3] = {p[0] - og[k][0], p[1] - og[k][1], p[2] - og[k][2]}, col[3];
      cross3(ax[k], dd, col);
      for (int r = 0; r < 3; r++) GLS(GL_JA + 21 * b + 7 * r + k) = col[0] * GRS(base + 3 + 3 * r) + col[1] * GRS(base + 4 + 3 * r) + col[2] * GRS(base + 5 + 3 * r);
    }
  }
  for (int j = 0; j < narm; j++) {
    const int base = GG_CON + (GEN_ARMSEG * GEN_SEG + j) * GREC;
    double p[3] = {GRS(base), GRS(base + 1), GRS(base + 2)};
    for (int k = 0; k < NARM; k++) {
      double dd[3] = {p[0] - og[k][0], p[1] - og[k][1], p[2] - og[k][2]}, col[3];
      cross3(ax[k], dd, col);
      for (int r = 0; r < 3; r++) GLS(GL_JA + 21 * (GEN_MAXNB + j) + 7 * r + k) = col[0] * GRS(base + 3 + 3 * r) + col[1] * GRS(base + 4 + 3 * r) + col[2] * GRS(base + 5 + 3 * r);
    }
  }
}
// ---- phase 4: islands.  Every lane derives the connected components of {cubes, arm} under cube-cube and rod contacts from the
// per-cube info words (identical result in all lanes), in the order of their first block, with their storage offsets.
#if defined(D3IL_HOST_STATS)
inline long g_isl_hist[80] = {0};      // host diagnostics: joint solves by island shape (cubes + 5 arm + 10 min(rod contacts, 3)); [40 + key]: their Newton iterations
#endif
struct IslSet { Isl isl[GEN_MAXNB + 1]; unsigned cpk[GEN_MAXNB + 1]; int first[GEN_MAXNB + 1]; bool fast[GEN_MAXNB + 1]; int n; };      // fast: a tree island, solved by gen_tree_solve
template <bool RS>
D3IL_HD void gen_islands(const GenConsts& gc_, const PushScratch sc, IslSet& out) {
  D3IL_GEN_CONSTS(gc_, gc);
  const int nb = gc.nb;
  unsigned adj[GEN_MAXNB + 1], cnt[GEN_MAXNB + 1];
#pragma unroll
  for (int b = 0; b <= GEN_MAXNB; b++) { adj[b] = 0; cnt[b] = 0; }
#pragma unroll
  for (int b = 0; b < GEN_MAXNB; b++) if (b < nb) {
    unsigned info = (unsigned)GLS(GL_INFO + b);
    unsigned m = (info >> 5) & 15u;
    if ((info >> 9) & 1) m |= 1u << nb;
    cnt[b] = info & 31u;
    adj[b] |= m;
#pragma unroll
    for (int d = 0; d <= GEN_MAXNB; d++) if ((m >> d) & 1) adj[d] |= 1u << b;
  }
  if (RS) cnt[nb] = (unsigned)GLS(GL_INFO + 8);      // rod <-> static contacts: they belong to the arm block
  unsigned seen = 0;
  int hoff = 0, voff = 0;
  out.n = 0;
#pragma unroll
  for (int b = 0; b <= GEN_MAXNB; b++) if (b <= nb && !((seen >> b) & 1)) {
    unsigned mask = 1u << b;
#pragma unroll
    for (int sweep = 0; sweep < GEN_MAXNB; sweep++)
#pragma unroll
      for (int d = 0; d <= GEN_MAXNB; d++) if ((mask >> d) & 1) mask |= adj[d];
    seen |= mask;
    Isl t = gen_island(mask, nb);
    t.hoff = hoff; t.voff = voff;
    unsigned cpk = 0;
    int k = 0;
#pragma unroll
    for (int d = 0; d <= GEN_MAXNB; d++) if ((mask >> d) & 1) { cpk |= cnt[d] << (5 * k); k++; }
    {
      unsigned cadj[GEN_MAXNB], rodm = 0;
#pragma unroll
      for (int d = 0; d < GEN_MAXNB; d++) { cadj[d] = adj[d] & ((1u << nb) - 1u); if ((adj[d] >> nb) & 1) rodm |= 1u << d; }
      const unsigned cm = mask & ((1u << nb) - 1u);
      out.fast[out.n] = cm != 0 && gt_island_fast(cadj, rodm, cm, nb, GLS(GL_TR) != 0.0);
    }
    out.isl[out.n] = t; out.cpk[out.n] = cpk; out.first[out.n] = b; out.n++;
    hoff += t.m * (t.m + 1) / 2; voff += t.m;
  }
}
D3IL_HD bool gen_uncoupled(const GenConsts& gc_, const PushScratch sc) {   // no cube-cube and no rod contact in this environment
  D3IL_GEN_CONSTS(gc_, gc);
  unsigned c = 0;
#pragma unroll
  for (int b = 0; b < GEN_MAXNB; b++) if (b < gc.nb) c |= (unsigned)GLS(GL_INFO + b) >> 5;
  return c == 0;
}
// The arm on its own with ONE rod <-> static box contact and no arm joint at a limit (the rod pressing on a wall of a gate): the situation of the
// Avoiding task's rod <-> obstacle contact, solved the same way - panda_step.h's register-resident Newton in the 5-dimensional constraint space
// (contact rows + the two finger-limit rows) on the group's lane 0 - instead of the four-lane island solver (every rod pressing on a wall with the
// cubes out of its way: 1.45 ms per step at 4096 environments against 1.40 at rest, tools/gpu_gen_rest_time.py).  Same optimum: both minimise the
// same strictly convex cost to the same tolerances.
D3IL_NOINLINE inline bool gen_arm_contact1(const GenConsts& gc_, const PushScratch sc) {
  D3IL_GEN_CONSTS(gc_, gc);
  const int arm0 = 6 * gc.nb, base = GG_CON + GEN_ARMSEG * GEN_SEG * GREC;
  double M[45], L[45], d[NDOF], id[NDOF], a0[NDOF], va[NDOF], fsv[NDOF];
#pragma unroll
  for (int i = 0; i < 45; i++) M[i] = GLS(GL_M + i);
#pragma unroll
  for (int k = 0; k < NDOF; k++) { a0[k] = GLS(GL_A0 + arm0 + k); va[k] = GLS(GL_VEL + arm0 + k); }
  bool ok = ldl9(M, L, d, id);
  symv9(M, a0, fsv);                                   // the smooth force M qacc_smooth, for the gradient scale
  double fn = 0, md = 0;
#pragma unroll
  for (int k = 0; k < NDOF; k++) { fn += fsv[k] * fsv[k]; md += M[tri(k, k)]; }
  const double gscale = (1.0 + sqrt(fn)) / (md / NDOF);
  RodContact rc;
  rc.active = true;
  const int set = gc.ns + 2 + (int)GRS(base + 14);
  const double dist = GRS(base + 12);
  double vel[3] = {0, 0, 0};
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int k = 0; k < NARM; k++) { rc.J[r][k] = GLS(GL_JA + 21 * GEN_MAXNB + 7 * r + k); vel[r] += rc.J[r][k] * va[k]; }
  const double imp = impedance(gc.ct_solimp[set], dist);
  const double Rn = fmax(1e-15, (1 - imp) / imp * gc.rod_invw), Rt = Rn / fmax(1e-15, gc.impratio), f0 = gc.ct_fric[set];
  rc.mu = f0 * sqrt(Rt / Rn); rc.fric[0] = f0; rc.fric[1] = f0;
  rc.D[0] = 1 / Rn; rc.D[1] = 1 / Rt; rc.D[2] = 1 / Rt;
  rc.aref[0] = -gc.ct_B[set] * vel[0] - gc.ct_K[set] * imp * dist;
  rc.aref[1] = -gc.ct_B[set] * vel[1]; rc.aref[2] = -gc.ct_B[set] * vel[2];
  double fsign[NFING], fD[NFING], faref[NFING], fc[NDOF], warm[6];
#pragma unroll
  for (int k = 0; k < NFING; k++) { fsign[k] = GLS(GL_LIM + 3 * (NARM + k)); fD[k] = GLS(GL_LIM + 3 * (NARM + k) + 1); faref[k] = GLS(GL_LIM + 3 * (NARM + k) + 2); }
  warm[5] = 0.0;
  ok = solve_contact5(L, id, a0, rc, fsign, fD, faref, gscale, fc, warm) && ok;
  ldl9_solve(L, id, fc);                               // qacc = qacc_smooth + M^-1 qfrc_constraint
#pragma unroll
  for (int k = 0; k < NDOF; k++) GLS(GL_X + arm0 + k) = a0[k] + fc[k];
  return ok;
}
// does gen_phase4_multi have anything to do for this environment?  (its own decisions; evaluated by the step kernel to send the WHOLE wave through the joint
// solver for the environments that need it, one after the other)
template <bool RS>
D3IL_HD bool gen_joint_work(const GenConsts& gc_, const PushScratch sc) {
  D3IL_GEN_CONSTS(gc_, gc);
  if (gen_uncoupled(gc, sc)) return !(GLS(GL_INFO + 4) == 0 && (RS ? (unsigned)GLS(GL_INFO + 8) : 0u) == 0);
  IslSet is;
  gen_islands<RS>(gc, sc, is);
  bool any = false;
#pragma unroll
  for (int k = 0; k <= GEN_MAXNB; k++) if (k < is.n) {
    const bool arm_alone = is.isl[k].n == 1 && is.isl[k].arm;
    if (is.fast[k]) continue;
    if (is.isl[k].n > 1 || (arm_alone && (GLS(GL_INFO + 4) != 0 || (RS && GLS(GL_INFO + 8) != 0)))) any = true;
  }
  return any;
}
// phase 4b (all nl lanes of the group together, lane l): the islands with more than one block, one after the other, and the arm
// on its own when one of its seven joints is at a limit or its rod is on a static box (otherwise it keeps the phase-1 solution)
template <bool RS>
D3IL_HD void gen_phase4_multi(const GenConsts& gc_, const PushScratch sc, int l, int nl, bool warm_valid, unsigned& fl) {
  D3IL_GEN_CONSTS(gc_, gc);
  if (gen_uncoupled(gc, sc)) {
    const unsigned narm = RS ? (unsigned)GLS(GL_INFO + 8) : 0u;
    if (GLS(GL_INFO + 4) == 0 && narm == 0) return;            // the usual case: nothing to do
    if (RS && GLS(GL_INFO + 4) == 0 && narm == 1) { if (l == 0 && !gen_arm_contact1(gc, sc)) fl |= F_SOLVER_FAIL; gen_sync(); return; }
    Isl isl = gen_island(1u << gc.nb, gc.nb);                   // only the arm, with a joint at its limit or the rod on a static box
    GEN_FOR_DOFS(ci, gi) GLS(GL_X + gi) = warm_valid ? GWARM(gi) : GLS(GL_A0 + gi);
    gen_sync();
    if (!gen_solve<RS>(gc, sc, isl, narm, l, nl)) fl |= F_SOLVER_FAIL;
    gen_sync();
    return;
  }
  IslSet is;
  gen_islands<RS>(gc, sc, is);
  // the k-th island that needs the joint solver: packed so that the groups of a wave run their k-th solves side by side
  unsigned todo = 0;
  int ntodo = 0;
#pragma unroll
  for (int k = 0; k <= GEN_MAXNB; k++) if (k < is.n) {
    const bool arm_alone = is.isl[k].n == 1 && is.isl[k].arm;
    if (is.fast[k]) continue;      // solved by the tree solver (gen_tree.h)
    if (is.isl[k].n > 1 || (arm_alone && (GLS(GL_INFO + 4) != 0 || (RS && GLS(GL_INFO + 8) != 0)))) { todo |= (unsigned)k << (4 * ntodo); ntodo++; }
  }
  for (int j = 0; j < ntodo; j++) {
    const int k = (int)((todo >> (4 * j)) & 15u);
    Isl isl = is.isl[0];
    unsigned cpk = is.cpk[0];
#pragma unroll
    for (int q = 1; q <= GEN_MAXNB; q++) if (q == k) { isl = is.isl[q]; cpk = is.cpk[q]; }
    if (RS && isl.n == 1 && isl.arm && GLS(GL_INFO + 4) == 0 && cpk == 1u) {      // the arm on its own, the rod on one static box
      if (l == 0 && !gen_arm_contact1(gc, sc)) fl |= F_SOLVER_FAIL;
      gen_sync();
      continue;
    }
    GEN_FOR_DOFS(ci, gi) GLS(GL_X + gi) = warm_valid ? GWARM(gi) : GLS(GL_A0 + gi);      // identical stores from every lane
    gen_sync();
#if defined(D3IL_HOST_STATS)
    int hkey_ = 0; long hit0_ = g_stats.newton_iters;
    { int nrod = 0; GEN_FOR_BLOCKS(b) if (b < gc.nb && (((unsigned)GLS(GL_INFO + b) >> 9) & 1)) nrod++;
      hkey_ = (isl.n - (isl.arm ? 1 : 0)) + (isl.arm ? 5 : 0) + 10 * (nrod > 3 ? 3 : nrod); g_isl_hist[hkey_]++;
      int nedge = 0; GEN_FOR_BLOCKS(b2) if (b2 < gc.nb) nedge += __builtin_popcount(((unsigned)GLS(GL_INFO + b2) >> 5) & 15u);      // cube <-> cube pairs in contact inside the island
      if (nedge < 4) g_isl_hist[36 + nedge]++; }
#endif
    if (!gen_solve<RS>(gc, sc, isl, cpk, l, nl)) fl |= F_SOLVER_FAIL;
#if defined(D3IL_HOST_STATS)
    g_isl_hist[40 + hkey_] += g_stats.newton_iters - hit0_;
#endif
    gen_sync();
  }
}
// ---- phase 5: integration.  Arm (lane 0): (M + h B) qacc = M x with B on the fingers; cube l: mj_Euler with quaternion integration
template <class C>
D3IL_HD void gen_phase5_arm(const C& c0, const GenConsts& gc_, EnvState& st, const PushScratch sc) {
  D3IL_GEN_CONSTS(gc_, gc);
  D3IL_REFRESH(c0, c);
  const double h = c.timestep;
  const int arm0 = 6 * gc.nb;
  double M[45], xa[NDOF], rhs[NDOF], L[45], d[NDOF], id[NDOF];
#pragma unroll
  for (int i = 0; i < 45; i++) M[i] = GLS(GL_M + i);
  const bool via_lambda = GLS(GL_TR) == 2.0;      // the arm's island went through the tree solver: x_a = a0 + M^-1 W' lambda, i.e. M x_a = M a0 + W' lambda
#pragma unroll
  for (int k = 0; k < NDOF; k++) xa[k] = GLS(via_lambda ? GL_A0 + arm0 + k : GL_X + arm0 + k);
  symv9(M, xa, rhs);
  if (!ldl9(M, L, d, id)) st.flags |= F_SOLVER_FAIL;
  if (via_lambda) {
    int rb = 0;
#pragma unroll
    for (int b = 0; b < GEN_MAXNB; b++) if (b < gc.nb && (((unsigned)GLS(GL_INFO + b) >> 9) & 1)) rb = b;
#pragma unroll
    for (int k = 0; k < NARM; k++) rhs[k] += GLS(GL_JA + 21 * rb + k) * GLS(GL_TR + 1) + GLS(GL_JA + 21 * rb + 7 + k) * GLS(GL_TR + 2) + GLS(GL_JA + 21 * rb + 14 + k) * GLS(GL_TR + 3);
#pragma unroll
    for (int f = 0; f < NFING; f++) { const double sg = GLS(GL_LIM + 3 * (NARM + f)); rhs[NARM + f] += (sg != 0 ? sg : 1.0) * GLS(GL_TR + 4 + f); }
#pragma unroll
    for (int k = 0; k < NDOF; k++) xa[k] = rhs[k];
    ldl9_solve(L, id, xa);
  }
#pragma unroll
  for (int k = 0; k < NDOF; k++) GWARM(arm0 + k) = xa[k];
  double hb0 = h * c.f_damping[0], hb1 = h * c.f_damping[1];
  double l87 = L[tri(8, 7)];
  double S11 = d[8] + l87 * l87 * d[7];
  double d7n = d[7] + hb0, i7 = rcpd(d7n);
  double l87n = l87 * d[7] * i7;
  double d8n = S11 + hb1 - l87n * l87n * d7n;
  d[7] = d7n; d[8] = d8n; id[7] = i7; id[8] = rcpd(d8n); L[tri(8, 7)] = l87n;
  ldl9_solve(L, id, rhs);
#pragma unroll
  for (int k = 0; k < NDOF; k++) { st.v[k] += h * rhs[k]; st.q[k] += h * st.v[k]; }
}
D3IL_HD void gen_phase5_cube(const GenConsts& gc_, const PushScratch sc, int b, double h) {
  D3IL_GEN_CONSTS(gc_, gc);
  double xb[6];
  BoxState bx;
#pragma unroll
  for (int k = 0; k < 6; k++) { xb[k] = GLS(GL_X + 6 * b + k); GWARM(6 * b + k) = xb[k]; }
#pragma unroll
  for (int k = 0; k < 3; k++) bx.pos[k] = GLS(GL_POS + 3 * b + k);
#pragma unroll
  for (int k = 0; k < 4; k++) bx.quat[k] = GBX(b, 3 + k);
#pragma unroll
  for (int k = 0; k < 6; k++) bx.vel[k] = GLS(GL_VEL + 6 * b + k);
  cube_integrate(bx, xb, h);
#pragma unroll
  for (int k = 0; k < 3; k++) GBX(b, k) = bx.pos[k];
#pragma unroll
  for (int k = 0; k < 4; k++) GBX(b, 3 + k) = bx.quat[k];
#pragma unroll
  for (int k = 0; k < 6; k++) GBX(b, 7 + k) = bx.vel[k];
}

// one physics sub-step (mj_step) of arm + cubes with the group's lanes run one after the other (host build, reset kernel)
template <bool RS, class C>
D3IL_HD void gen_physics_substep_t(const C& c0, const GenConsts& gc_, EnvState& st, const PushScratch sc, const double* tau, const double* ffing) {
  D3IL_GEN_CONSTS(gc_, gc);
  D3IL_REFRESH(c0, c);
  const double grav[3] = {c.gravity[0], c.gravity[1], c.gravity[2]};
  const bool warm_valid = (st.flags & PF_WARM_VALID) != 0;
  unsigned fl = 0;
  int cnt[GEN_MAXNB];
  gen_phase1(c0, gc, st, sc, tau, ffing);
  for (int l = 0; l < gc.nb; l++) cnt[l] = gen_phase2(gc, sc, l, grav, fl);
  for (int l = 0; l < gc.nb; l++) gen_phase3(gc, sc, l, cnt[l], c.rod_r, c.rod_h, fl);
  if (RS && gc.rod_static) for (int l = 0; l < gc.nb; l++) gen_phase3r(gc, sc, l, gc.nb, c.rod_r, c.rod_h, fl);
  gen_phase3b<RS>(c0, gc, st, sc, gc.nb, fl);
  gen_arm_reduce<RS>(gc, sc, warm_valid);
  if (gen_uncoupled(gc, sc)) { for (int l = 0; l < gc.nb; l++) fl |= gen_lone_solve<1>(gc, sc, l, warm_valid, 0); }      // (the rule of the step kernel, gen_kernels.h)
  else fl |= gen_tree_solve<GEN_MAXNB>(gc, sc, 0, warm_valid);
  gen_phase4_multi<RS>(gc, sc, 0, 1, warm_valid, fl);
  gen_phase5_arm(c0, gc, st, sc);
  for (int l = 0; l < gc.nb; l++) gen_phase5_cube(gc, sc, l, c.timestep);
  st.flags |= fl | PF_WARM_VALID;
}
template <class C>
D3IL_HD void gen_physics_substep(const C& c0, const GenConsts& gc_, EnvState& st, const PushScratch sc, const double* tau, const double* ffing) {
  D3IL_GEN_CONSTS(gc_, gc);
  if (gc.rod_static) gen_physics_substep_t<true>(c0, gc, st, sc, tau, ffing);
  else gen_physics_substep_t<false>(c0, gc, st, sc, tau, ffing);
}

// ------------------------------------------------------------------------------------------------ Sorting task (sorting.py)
// Task state of Sorting_Env in two words: word 0 = mode[6] (2 bits each: value + 1) | mode_step << 12;  word 1 = min_inds[6]
// (3 bits each).  sorting.py:405-411 (reset), :460-507 (check_mode)
constexpr int GEN_SORT_OBS = 20;      // 2 + 3 * 6 at most (num_boxes = 6); Sorting-4 uses 14
D3IL_HD void sort_collect(const GenConsts& gc_, const PushScratch sc, double (*box)[7]) {
  D3IL_GEN_CONSTS(gc_, gc);
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
D3IL_HD bool sort_obs_success(const GenConsts& gc_, const double (*box)[7], const double* tcp, float* obs) {
  D3IL_GEN_CONSTS(gc_, gc);
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
D3IL_HD int sort_check_mode(const GenConsts& gc_, unsigned* task, const double (*box)[7]) {
  D3IL_GEN_CONSTS(gc_, gc);
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

// ------------------------------------------------------------------------------------------------ Inserting task (gate_insertion.py)
// Gate_Insertion_Env on the same engine (three 5 cm cubes, seventeen static walls, the rod also meets the walls).  Task state: word 0 = number of
// letters in `modes` | letter k (1 r, 2 g, 3 b) << (2 + 2 k); word 1 = mean_distance of the last step (a double, read back by the env class).
// get_observation (gate_insertion.py:278-309), _check_early_termination (:475-498), check_mean_dist (:434-446): distances over the full 3-D positions
D3IL_HD bool ins_obs_success(const GenConsts& gc_, const PushScratch sc, const double* tcp, float* obs, double* d3, double* mean_dist) {
  D3IL_GEN_CONSTS(gc_, gc);
  obs[0] = (float)tcp[0]; obs[1] = (float)tcp[1];
  bool ok = true;
  double sum = 0;
  for (int i = 0; i < 3; i++) {
    double b[7];
    for (int j = 0; j < 7; j++) b[j] = GBX(i, j);
    obs[2 + 3 * i] = (float)b[0]; obs[3 + 3 * i] = (float)b[1]; obs[4 + 3 * i] = (float)push_tan_yaw(b + 3);
    const double dx = b[0] - gc.ins_target[3 * i], dy = b[1] - gc.ins_target[3 * i + 1], dz = b[2] - gc.ins_target[3 * i + 2];
    d3[i] = sqrt(dx * dx + dy * dy + dz * dz);
    ok = ok && d3[i] <= gc.ins_min_dist;
    sum += d3[i];
  }
  *mean_dist = sum / 3;
  return ok;
}
// check_mode (:411-432) + step's info (:386-409): code = mode_dict[letters] (rgb 1, rbg 2, grb 3, gbr 4, brg 5, bgr 6) once all three letters are in, else 0;
// reported as code | number of letters << 3 (one / two / three_box_success are that number >= 1 / 2 / 3)
D3IL_HD int ins_check_mode(const GenConsts& gc_, unsigned* task, const double* d3) {
  D3IL_GEN_CONSTS(gc_, gc);
  unsigned n = task[0] & 3u;
  for (unsigned i = 0; i < 3; i++) {
    bool have = false;
    for (unsigned k = 0; k < n; k++) have = have || ((task[0] >> (2 + 2 * k)) & 3u) == i + 1;
    if (d3[i] <= gc.ins_min_dist && !have) { task[0] |= (i + 1) << (2 + 2 * n); n++; }
  }
  task[0] = (task[0] & ~3u) | n;
  int code = 0;
  if (n == 3) {
    const unsigned a = (task[0] >> 2) & 3u, b = (task[0] >> 4) & 3u;      // first and second letter decide
    code = a == 1 ? (b == 2 ? 1 : 2) : (a == 2 ? (b == 1 ? 3 : 4) : (b == 1 ? 5 : 6));
  }
  return code | (int)(n << 3);
}

// ------------------------------------------------------------------------------------------------ Pushing task (pushing.py) on this engine
// Block_Push_Env with its two cubes as the engine's cubes 0 (red) and 1 (green), the table slabs and the frame beams as static boxes.  Task state as in
// push_step.h: first_visit + 1 and mode + 1 in the low flag bits (PF_FIRST_MASK / PF_MODE_MASK); the two task rows of the state buffer carry the step's
// info['mean_distance'] and reward (they ARE the info_f64 rows of the boundary).  Same arithmetic as push_dists / push_step_begin / push_step_end.
D3IL_HD void gpush_dists(const GenConsts& gc_, const PushScratch sc, double* d) {   // rr rg gr gg
  D3IL_GEN_CONSTS(gc_, gc);
  for (int b = 0; b < 2; b++) for (int t = 0; t < 2; t++) {
    const double dx = GBX(b, 0) - gc.ins_target[3 * t], dy = GBX(b, 1) - gc.ins_target[3 * t + 1], dz = GBX(b, 2) - gc.ins_target[3 * t + 2];
    d[2 * b + t] = sqrt(dx * dx + dy * dy + dz * dz);
  }
}
D3IL_HD void gpush_obs(const PushScratch sc, const double* tcp, float* obs) {   // pushing.py:255-280
  obs[0] = (float)tcp[0]; obs[1] = (float)tcp[1];
  for (int b = 0; b < 2; b++) {
    double q[4] = {GBX(b, 3), GBX(b, 4), GBX(b, 5), GBX(b, 6)};
    obs[2 + 3 * b] = (float)GBX(b, 0); obs[3 + 3 * b] = (float)GBX(b, 1); obs[4 + 3 * b] = (float)push_tan_yaw(q);
  }
}
// before the physics: obs, reward (get_reward, pushing.py:379-407), done (gym_env_wrapper.py:88-90,124-137)
D3IL_HD bool gpush_step_begin(const GenConsts& gc_, const EnvState& st, const PushScratch sc, float* obs) {
  D3IL_GEN_CONSTS(gc_, gc);
  gpush_obs(sc, st.tcp, obs);
  double d[4]; gpush_dists(gc, sc, d);
  const double dx = st.tcp[0] - GBX(0, 0), dy = st.tcp[1] - GBX(0, 1);
  GTASK(1) = -(sqrt(dx * dx + dy * dy) + d[0]);
  const double md = gc.ins_min_dist;
  return (d[0] <= md && d[3] <= md) || (d[1] <= md && d[2] <= md);      // push_success, pushing.py:440-459
}
// after the physics: success, first-visit mode logic (pushing.py:335-377); returns the mode (-1 .. 3)
D3IL_HD int gpush_step_end(const GenConsts& gc_, EnvState& st, const PushScratch sc) {
  D3IL_GEN_CONSTS(gc_, gc);
  double d[4]; gpush_dists(gc, sc, d);
  const double md = gc.ins_min_dist;
  const bool succ = (d[0] <= md && d[3] <= md) || (d[1] <= md && d[2] <= md);
  st.flags &= ~F_SUCCESS;
  if (succ) st.flags |= F_SUCCESS | F_TERMINATED;
  int first = (int)(st.flags & PF_FIRST_MASK) - 1, visit = -1, mode = -1;
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
  st.flags = (st.flags & ~(PF_FIRST_MASK | PF_MODE_MASK)) | (unsigned)(first + 1) | ((unsigned)(mode + 1) << PF_MODE_SHIFT);
  GTASK(0) = 0.5 * (fmin(d[0], d[1]) + fmin(d[2], d[3]));
  return mode;
}

// ------------------------------------------------------------------------------------------------ env level
template <class C>
D3IL_HD void gen_control_and_physics(const C& c, const GenConsts& gc_, EnvState& st, const PushScratch sc, const double* q_des, const double* qd_des,
                                     double set_width, bool grasp) {
  D3IL_GEN_CONSTS(gc_, gc);
  double tau[NARM], ff[NFING];
  push_control(c, st, q_des, qd_des, set_width, grasp, tau, ff);
  gen_physics_substep(c, gc, st, sc, tau, ff);
}
// Sorting_Env.reset(random=False, context) (sorting.py:545-575): ctx = nb x (pos3, quat4) in the order red_1.., blue_1..
template <class C>
D3IL_HD void gen_env_reset(const C& c, const GenConsts& gc_, EnvState& st, const PushScratch sc, const double* init_qpos, const double* ctx, float* obs) {
  D3IL_GEN_CONSTS(gc_, gc);
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
  if (gc.task == GEN_TASK_INSERTING) { double d3[3], md; ins_obs_success(gc, sc, st.tcp, obs, d3, &md); return; }
  if (gc.task == GEN_TASK_PUSHING) { gpush_obs(sc, st.tcp, obs); return; }
  double box[6][7];
  sort_collect(gc, sc, box);
  sort_obs_success(gc, box, st.tcp, obs);
}
// before the physics of a step: observation and done (gym_env_wrapper.py:88-90,124-137)
D3IL_HD void sort_step_begin(const GenConsts& gc_, EnvState& st, const PushScratch sc, float* obs, unsigned char* done, int max_steps) {
  D3IL_GEN_CONSTS(gc_, gc);
  bool succ;
  if (gc.task == GEN_TASK_INSERTING) { double d3[3], md; succ = ins_obs_success(gc, sc, st.tcp, obs, d3, &md); }
  else if (gc.task == GEN_TASK_PUSHING) succ = gpush_step_begin(gc, st, sc, obs);
  else {
    double box[6][7];
    sort_collect(gc, sc, box);
    succ = sort_obs_success(gc, box, st.tcp, obs);
  }
  bool fin = (st.flags & F_TERMINATED) != 0;
  if (!fin && succ) { st.flags |= F_TERMINATED; fin = true; }
  if (!fin && st.step >= max_steps - 1) fin = true;
  *done = fin ? 1 : 0;
}
// after the physics: success and the completion-order mode code (sorting.py:444-458)
D3IL_HD void sort_step_end(const GenConsts& gc_, EnvState& st, const PushScratch sc, int* mode_code) {
  D3IL_GEN_CONSTS(gc_, gc);
  st.step++;
  float dummy[GEN_SORT_OBS];
  if (gc.task == GEN_TASK_PUSHING) { *mode_code = gpush_step_end(gc, st, sc); return; }
  if (gc.task == GEN_TASK_INSERTING) {
    double d3[3], md;
    const bool succ = ins_obs_success(gc, sc, st.tcp, dummy, d3, &md);
    st.flags &= ~F_SUCCESS;
    if (succ) st.flags |= F_SUCCESS | F_TERMINATED;
    unsigned task[1] = {(unsigned)GTASK(0)};
    *mode_code = ins_check_mode(gc, task, d3);
    GTASK(0) = (double)task[0]; GTASK(1) = md;
    return;
  }
  double box[6][7];
  sort_collect(gc, sc, box);
  bool succ = sort_obs_success(gc, box, st.tcp, dummy);
  st.flags &= ~F_SUCCESS;
  if (succ) st.flags |= F_SUCCESS | F_TERMINATED;
  unsigned task[2] = {(unsigned)GTASK(0), (unsigned)GTASK(1)};
  *mode_code = sort_check_mode(gc, task, box);
  GTASK(0) = (double)task[0]; GTASK(1) = (double)task[1];
}
template <bool FAST, class C>
D3IL_HD void gen_env_step(const C& c, const GenConsts& gc_, EnvState& st, const PushScratch sc, const double* action, float* obs, unsigned char* done,
                          int* mode_code, int n_substeps, int max_steps) {
  D3IL_GEN_CONSTS(gc_, gc);
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
  // Static boxes a cube can meet: box geoms on joint-less body chains whose collision bits match the cubes' and whose bounding box
  // reaches the region above the table top (grown by the cube's circumradius).  Two groups: (i) what stands INSIDE the table_plane
  // footprint shrunk by 8 cm - in the Sorting scene table_plane, support_body, the eight bin walls and the platform -, tested every
  // sub-step; (ii) the aluminium profiles of the frame whose tops form a 19 mm rim around the table top (lab_surrounding.xml:3-114:
  // front / back / side upper beams and the four posts), tested only for a cube whose centre has left the shrunk footprint.  A cube
  // centre beyond the outer faces of the frame raises PF_OFF_TABLE (it would drop to the floor plane, which this engine does not model).
  const double rcirc = std::sqrt(gc.box_half[0] * gc.box_half[0] + gc.box_half[1] * gc.box_half[1] + gc.box_half[2] * gc.box_half[2]);
  int ns = 0, table = -1;
  double wlo[3] = {0, 0, 0}, whi[3] = {0, 0, 0}, olo[2] = {0, 0}, ohi[2] = {0, 0};
  bool taken[D3IL_MAXGEOM];
  int st_geom[GEN_MAXNS];
  for (int g = 0; g < D3IL_MAXGEOM; g++) taken[g] = false;
  for (int pass = 0; pass < 3; pass++) {
  for (int g = 0; g < m.ngeom; g++) {
    if (m.geom_type[g] != D3IL_GEOM_BOX || taken[g]) continue;
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
    double ext[3];
    for (int i = 0; i < 3; i++) ext[i] = std::fabs(R[3 * i]) * m.geom_size[g][0] + std::fabs(R[3 * i + 1]) * m.geom_size[g][1] + std::fabs(R[3 * i + 2]) * m.geom_size[g][2];
    if (pass == 0) {   // the table top: the 0.49 x 0.98 x 0.001 slab (lab_surrounding.xml:3-4)
      if (std::fabs(m.geom_size[g][0] - 0.49) < 1e-12 && std::fabs(m.geom_size[g][1] - 0.98) < 1e-12 && std::fabs(m.geom_size[g][2] - 0.001) < 1e-12) {
        table = g;
        for (int i = 0; i < 2; i++) {
          wlo[i] = p[i] - ext[i] + 0.08; whi[i] = p[i] + ext[i] - 0.08; gc.in_lo[i] = wlo[i]; gc.in_hi[i] = whi[i]; wlo[i] -= rcirc; whi[i] += rcirc;
          olo[i] = p[i] - ext[i] - rcirc; ohi[i] = p[i] + ext[i] + rcirc;      // anything a cube ON the table can touch
          gc.ws_lo[i] = p[i] - ext[i]; gc.ws_hi[i] = p[i] + ext[i];            // grown below to the outer faces of the frame
        }
        wlo[2] = p[2] + ext[2] - rcirc;
      }
      continue;
    }
    if (table < 0) { *err = "table slab not found"; return -1; }
    if (pass == 1) { if (p[0] + ext[0] < wlo[0] || p[0] - ext[0] > whi[0] || p[1] + ext[1] < wlo[1] || p[1] - ext[1] > whi[1] || p[2] + ext[2] < wlo[2]) continue; }
    else if (p[0] + ext[0] < olo[0] || p[0] - ext[0] > ohi[0] || p[1] + ext[1] < olo[1] || p[1] - ext[1] > ohi[1] || p[2] + ext[2] < wlo[2]) continue;
    if (m.geom_margin[g] != 0 || m.geom_gap[g] != 0) { *err = "static boxes with a contact margin are not supported"; return -1; }
    if (ns >= GEN_MAXNS) { *err = "too many static boxes"; return -1; }
    for (int k = 0; k < 3; k++) { gc.st_c[ns][k] = p[k]; gc.st_h[ns][k] = m.geom_size[g][k]; }
    for (int k = 0; k < 9; k++) gc.st_R[ns][k] = R[k];
    gc.st_first[ns] = g < g0 ? 1 : 0;
    st_geom[ns] = g;
    mix(g, g0, ns);
    taken[g] = true;
    if (pass == 2) for (int i = 0; i < 2; i++) { gc.ws_lo[i] = std::fmin(gc.ws_lo[i], p[i] - ext[i]); gc.ws_hi[i] = std::fmax(gc.ws_hi[i], p[i] + ext[i]); }
    ns++;
  }
  if (pass == 1) gc.ns_core = ns;
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
  gc.task = m.task_id == D3IL_TASK_INSERTING ? GEN_TASK_INSERTING : (m.task_id == D3IL_TASK_PUSHING ? GEN_TASK_PUSHING : GEN_TASK_SORTING);
  if (gc.task == GEN_TASK_PUSHING) {
    if (gc.nb != 2) { *err = "the Pushing task has two cubes"; return -1; }
    for (int k = 0; k < 6; k++) gc.ins_target[k] = m.task_f[k];
    gc.ins_min_dist = m.task_f[6];
  }
  if (gc.task == GEN_TASK_INSERTING) {
    if (gc.nb != 3) { *err = "the Inserting task has three push boxes"; return -1; }
    for (int k = 0; k < 9; k++) gc.ins_target[k] = m.task_f[k];
    gc.ins_min_dist = m.task_f[9];
    // the rod works between the walls of the gates: rod <-> static box pairs (the oracle evaluates them for this task, oracle/d3il_oracle.c pair_supported)
    gc.rod_static = 1;
    if (m.geom_margin[m.rod_geom] != 0 || m.geom_gap[m.rod_geom] != 0) { *err = "a rod with a contact margin is not supported"; return -1; }
    for (int s = 0; s < ns; s++) {
      const int g = st_geom[s];
      if (g > m.rod_geom) { *err = "unexpected geom order (static box after the rod)"; return -1; }
      gc.st_rod[s] = ((m.geom_contype[g] & m.geom_conaffinity[m.rod_geom]) || (m.geom_contype[m.rod_geom] & m.geom_conaffinity[g])) ? 1 : 0;
      mix(g, m.rod_geom, ns + 2 + s);
    }
  }
  for (int k = 0; k < 3; k++) gc.absent[k] = m.body_pos[m.nbody - 1][k];
  for (int k = 0; k < 4; k++) gc.absent[3 + k] = m.body_quat[m.nbody - 1][k];
  gc.impratio = m.impratio;
  gc.rod_invw = pcst.rod_invweight0;
  return 0;
}

}  // namespace d3il
