/* d3il_rollout.h - C ABI of libd3il_rollout.so, the MI355X-native batched replacement for the D3IL
 * evaluation env.step() path.  Plain C, plain pointers and sizes; no torch types.
 *
 * The reference has no FFI layer: its operator boundary is the Gym-style Python protocol of the task
 * envs.  Each entry point below names the reference interface it replaces (paths relative to
 * /root/reference); the Python mirror of those classes lives in d3il_amd/envs and d3il_amd/simulation.
 *
 * Conventions
 *  - every function returns 0 on success, a negative D3IL_E* code on failure; d3il_last_error() gives
 *    the message of the calling thread's last failure.
 *  - the library owns all device memory it hands out through d3il_get_buffers(); the caller owns the
 *    action buffer.  Device pointers are valid until d3il_destroy().
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Kernels are only enqueued;
 *    the caller synchronises (PyTorch: pass torch.cuda.current_stream().cuda_stream so that
 *    agent.predict() consumes observations in place without extra synchronisation).
 *  - one host thread per handle; calls on one handle are not re-entrant.
 *  - there is NO CPU fallback: d3il_create fails with D3IL_ENODEVICE when no HIP device is usable.
 *
 * State layout (device, f64): structure-of-arrays [D3IL_STATE_F64][stride], stride = n_envs rounded up
 * to 64, field order D3IL_STATE_*; one lane owns one environment, loads/stores are coalesced.
 */
#ifndef D3IL_ROLLOUT_H
#define D3IL_ROLLOUT_H
#include <stddef.h>
#include <stdint.h>
#include "d3il_model_blob.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct d3il_handle_s* d3il_handle;

enum {
  D3IL_OK = 0, D3IL_EINVAL = -1, D3IL_EBLOB = -2, D3IL_ENODEVICE = -3, D3IL_EHIP = -4, D3IL_EUNSUPPORTED = -5,
  D3IL_ESTATE = -6, D3IL_ERCCL = -7
};

/* Tasks (task_id of d3il_create; the enum itself is D3IL_TASK_* in d3il_model_blob.h: 0 Avoiding, 1 Pushing, 2 Sorting, 3 Stacking, 4 Aligning, 5 Inserting) and
 * their shapes:
 *   task      reference env (environments/d3il/envs/...)            action (device f64, row-major)              obs f32   contexts f64     state rows
 *   Avoiding  gym_avoiding/envs/avoiding.py ObstacleAvoidanceEnv    [n][7] desired TCP x y z qw qx qy qz        [n][2]    none             42
 *   Pushing   gym_pushing/envs/pushing.py Block_Push_Env            [n][7] same                                 [n][8]    [n][14]          91
 *   Sorting   gym_sorting/envs/sorting.py Sorting_Env (2 / 4 boxes) [n][7] same                                 [n][2+3b] [n][7 b]         42+13b+(9+6b)+2
 *   Stacking  gym_stacking/envs/stacking.py CubeStacking_Env        [n][8] 7 desired joint positions + gripper   [n][12]   [n][21]          94
 *                                                                   command (open iff > 0.075, stacking.py:337-346)
 *   Aligning  gym_aligning/envs/aligning.py Robot_Push_Env          [n][7] as Avoiding (the harness commands z)  [n][17]   [n][14]          77
 *   Inserting gym_inserting/envs/gate_insertion.py Gate_Insertion_Env [n][7] as Avoiding                          [n][11]   [n][21]          110
 */

/* f64 state fields per environment, in SoA order */
enum {
  D3IL_STATE_QPOS = 0,    /* 9: 7 arm joints, 2 fingers */
  D3IL_STATE_QVEL = 9,    /* 9 */
  D3IL_STATE_BIAS = 18,   /* 7: qfrc_bias of the last forward pass (gravity compensation is one sub-step stale) */
  D3IL_STATE_TCP = 25,    /* 3: TCP xpos of the last forward pass (what MjRobot.receiveState reads) */
  D3IL_STATE_IK_Q = 28,   /* 7: CartPosQuatImpedenceController.old_q */
  D3IL_STATE_IK_QD = 35,  /* 7: CartPosQuatImpedenceController.old_des_joint_vel */
  D3IL_STATE_F64 = 42,
  /* Pushing appends: per cube pos[3] quat[4] vel[6] (qpos/qvel of its free joint: linear velocity in world axes, angular
   * velocity in body axes), then the constraint solver's warm start qacc[21] (cube1, cube2, arm), then the two task rows info['mean_distance'] and reward
   * (pushing.py:335-407) - d3il_buffers.info_f64 points AT these two rows (a view of the state buffer, no copy).  91 rows since the task runs on the generic
   * engine (library version 2; 89 before: a version-1 checkpoint lacks the task rows).  Size state buffers from d3il_buffers.state_rows, not from a constant. */
  D3IL_PUSH_STATE_BOX = 42, D3IL_PUSH_STATE_WARM = 68, D3IL_PUSH_STATE_TASK = 89, D3IL_PUSH_STATE_F64 = 91,
  /* Sorting-4 (sorting.py): per cube pos[3] quat[4] vel[6] in the order red_1, red_2, blue_1, blue_2, the solver's warm start
   * qacc[33] (cubes, arm), then the task state of Sorting_Env as two words stored as doubles: word 0 = mode[6], two bits each
   * (value + 1), | mode_step << 12;  word 1 = min_inds[6], three bits each (sorting.py:405-411, 460-507) */
  D3IL_SORT_STATE_BOX = 42, D3IL_SORT_STATE_WARM = 94, D3IL_SORT_STATE_TASK = 127, D3IL_SORT_STATE_F64 = 129,
  /* Stacking (stacking.py; robot panda_invisible.xml, joint-space controller: no IK rows): rows 0..27 as above up to the TCP (qpos 9, qvel 9,
   * qfrc_bias 7, TCP 3), then per box pos[3] quat[4] vel[6] in the order red, green, blue (D3IL_STACK_STATE_BOX + 13 k), then the
   * constraint solver's warm start qacc[27] (box 0, box 1, box 2, arm 9).  The task state of CubeStacking_Env (order in which the boxes
   * reached the target zone, stacking.py:395-419) lives in the flag word (D3IL_SFLAG_*). */
  D3IL_STACK_STATE_BOX = 28, D3IL_STACK_STATE_WARM = 67, D3IL_STACK_STATE_F64 = 94,
  /* Aligning (gym_aligning/envs/aligning.py; the rod robot, Cartesian controller: rows 0..41 as for Avoiding): the free compound body
   * (robot_push_box.xml: plate + four walls) pos[3] quat[4] vel[6] as MuJoCo's free joint holds them (body origin, body-frame angular velocity),
   * the solver's warm start qacc[15] (box 6 - in centre-of-mass coordinates -, arm 9), the target pose pos[3] quat[4] of the context
   * (aligning.py:107-122; it only enters observation, reward and success).  Flag word: the Pushing bits (mode + 1 in D3IL_PFLAG_MODE_MASK: 0 / 1 =
   * rod inside / outside the walls, aligning.py:288-312; WARM_VALID, OFF_TABLE, CON_OVERFLOW) and D3IL_SFLAG_HAND_NEAR. */
  D3IL_ALIGN_STATE_BOX = 42, D3IL_ALIGN_STATE_WARM = 55, D3IL_ALIGN_STATE_TARGET = 70, D3IL_ALIGN_STATE_F64 = 77,
  /* Inserting (gate_insertion.py; the Sorting layout with three cubes push_box1..3 = red, green, blue): cubes 13 each, warm start qacc[27], then the task
   * state of Gate_Insertion_Env: word 0 = number of letters in `modes` | letter k (1 r, 2 g, 3 b) << (2 + 2 k) (gate_insertion.py:411-432); word 1 =
   * info['mean_distance'] of the last step, a double (:434-446).  mode buffer: info['mode'] (mode_dict code once all three letters are in, else 0,
   * :386-409) | number of letters << 3 (one / two / three_box_success = that number >= 1 / 2 / 3).  Contexts: 3 x (x, y, z = 0, quat) (:94-113). */
  D3IL_INS_STATE_BOX = 42, D3IL_INS_STATE_WARM = 81, D3IL_INS_STATE_TASK = 108, D3IL_INS_STATE_F64 = 110
};
/* bits of the per-environment u32 flag word */
enum {
  D3IL_FLAG_MODE_MASK = 0x1FF,       /* 9 sticky mode bits, avoiding.py:173-202 */
  D3IL_FLAG_L1 = 1 << 9, D3IL_FLAG_L2 = 1 << 10, D3IL_FLAG_L3 = 1 << 11,
  D3IL_FLAG_TERMINATED = 1 << 12, D3IL_FLAG_SUCCESS = 1 << 13, D3IL_FLAG_ROD_CONTACT = 1 << 14,
  D3IL_FLAG_IK_VALID = 1 << 15,
  D3IL_FLAG_SOLVER_FAIL = 1 << 16,   /* a solver did not converge, or the action held NaN / Inf (the step then ran on a fixed safe set-point and the episode is terminated) */
  D3IL_FLAG_MULTI_CONTACT = 1 << 17,
  /* Pushing reuses TERMINATED / SUCCESS / IK_VALID / SOLVER_FAIL and replaces the low bits: */
  D3IL_PFLAG_FIRST_MASK = 0x7,       /* first_visit + 1 (pushing.py:341-377) */
  D3IL_PFLAG_MODE_MASK = 0x38,       /* (mode + 1) << 3 */
  D3IL_PFLAG_WARM_VALID = 1 << 6,
  D3IL_PFLAG_CON_OVERFLOW = 1 << 18, /* more contacts than the solver holds (24) in some sub-step */
  D3IL_PFLAG_OFF_TABLE = 1 << 19,    /* a cube left the modelled part of the table */
  /* Stacking reuses TERMINATED / SUCCESS / SOLVER_FAIL / CON_OVERFLOW (more than 32 contacts) / OFF_TABLE and replaces the low bits: */
  D3IL_SFLAG_MODE_MASK = 0xFF,       /* order code n | c0 << 2 | c1 << 4 | c2 << 6: n boxes have reached the target zone, c_i = colour (0 r, 1 g, 2 b) of the
                                        i-th one (info['mode'] = "rgb"[c0] + ... , stacking.py:395-419); also what buf.mode holds */
  D3IL_SFLAG_WARM_VALID = 1 << 8,    /* the warm-start rows hold the accelerations of the previous sub-step (the gripper command is re-derived from the action every step, so
                                        RobotBase.grasp_flag needs no bit) */
  D3IL_SFLAG_HAND_NEAR = 1 << 20     /* a box came within the bounding box of a robot collision geom this engine does not evaluate */
};

typedef struct d3il_buffers {
  int32_t n_envs, stride, obs_dim, action_dim;
  float* obs;            /* [n_envs][obs_dim] f32, what get_observation() returns (avoiding.py:117-119, pushing.py:255-280) */
  uint8_t* done;         /* [n_envs] result of is_finished() of the last step (gym_env_wrapper.py:124-137) */
  uint8_t* success;      /* [n_envs] info[1] (avoiding.py:171) */
  uint16_t* mode;        /* [n_envs] Avoiding: 9-bit mode encoding, bit i = mode_encoding[i] (info[0]); Pushing: info['mode'] as int16 (-1..3);
                            Sorting: int(np.packbits(mode)[0]); Stacking: order code (D3IL_SFLAG_MODE_MASK) */
  double* state;         /* [state_rows][stride] */
  uint32_t* flags;       /* [stride] */
  int32_t* step_count;   /* [stride] env_step_counter */
  double* policy_des;    /* [3][stride] random-policy harness state: desired x, y and fixed z */
  double* info_f64;      /* [n_info_f64][stride] extra f64 step outputs; Pushing: info['mean_distance'], reward (pushing.py:335-407); Stacking: info['mean_distance'] */
  int32_t n_info_f64, state_rows;   /* state_rows: f64 state fields per environment (42 Avoiding, 91 Pushing, 129 Sorting-4, 94 Stacking, 77 Aligning, 110 Inserting) */
  uint8_t* last_reset;   /* [n_envs] environments reset by the last d3il_auto_reset (non-zero): per-lane harness / agent state re-latches from it */
} d3il_buffers;

/* Replaces: env construction + scene.start() (avoiding.py:52-92, core/Scene.py:95-108,
 * mj_scene_parser.py:36-53 building MjModel).  model_blob is a d3il_model_blob. */
int d3il_create(int task_id, int n_envs, int device_id, const void* model_blob, size_t blob_len, d3il_handle* out);
int d3il_destroy(d3il_handle h);

/* Replaces the part of env.start() that survives it: robot.init_qpos (avoiding.py:121-166).
 * The offline IK that produces init_qpos is host-side, cold (d3il_amd/controllers/offline_ik.py). */
int d3il_start(d3il_handle h, const double* init_qpos7);

/* Replaces env.reset() (avoiding.py:248-262).  env_mask: device u8[n_envs] (non-zero = reset that env) or
 * NULL for all.  contexts: NULL for Avoiding; Pushing (pushing.py:461-483 with random=False): device f64 [n_envs][14] =
 * per env 2 x (x, y, z, qw, qx, qy, qz) written into the cubes' qpos as BlockContextManager.set_context does (z = 0);
 * Sorting-4 (sorting.py:545-575): device f64 [n_envs][28] = 4 x (x, y, z = 0.05, quat), red boxes first (sorting.py:121-187).
 * Sorting outputs: obs f32 [n_envs][14] (TCP xy, then x, y, tan(yaw) per box), mode = int(np.packbits(mode[:4])[0]).
 * Stacking (stacking.py:449-481 with random=False): device f64 [n_envs][21] = 3 x (x, y, z = 0, qw, qx, qy, qz) for the red, green and blue box
 * (BlockContextManager.set_context, stacking.py:99-125); fingers opened before the reset sub-step (:474).  Stacking outputs: obs f32 [n_envs][12] =
 * (x, y, z, tan(yaw)) per box (stacking.py:228-277), mode = the order code of D3IL_SFLAG_MODE_MASK, info_f64[0] = info['mean_distance']. */
int d3il_reset(d3il_handle h, const uint8_t* env_mask, const double* contexts, void* stream);

/* Replaces env.step(action) (avoiding.py:168-171 over gym_env_wrapper.py:45-100): n_substeps fused physics
 * sub-steps.  actions: device f64[n_envs][7] = desired TCP (x, y, z, qw, qx, qy, qz), the array the harness
 * builds at avoiding_sim.py:64-66.  Stacking (stacking.py:331-393, 30 sub-steps): device f64[n_envs][8] = 7 desired joint positions of the
 * joint-space PD law + the gripper command (open iff > 0.075, else close_fingers), the array the harness builds at stacking_sim.py:99-106. */
int d3il_step(d3il_handle h, const double* actions, void* stream);

int d3il_get_buffers(d3il_handle h, d3il_buffers* out);

/* Golden replay / checkpointing: copies the SoA state + flags + step counters to/from host memory
 * (state: f64[state_rows][n_envs] packed with stride n_envs; flags u32[n_envs]; steps i32[n_envs]; any of the three may be NULL).
 * state_rows = the row count the CALLER's buffer was sized for; it must equal d3il_buffers.state_rows of the handle, otherwise nothing is copied
 * and D3IL_EINVAL is returned (a buffer sized from a stale constant cannot be overrun, a checkpoint of another layout cannot be loaded). */
int d3il_get_state(d3il_handle h, double* state, int32_t state_rows, uint32_t* flags, int32_t* steps);
int d3il_set_state(d3il_handle h, const double* state, int32_t state_rows, const uint32_t* flags, const int32_t* steps);

/* Device-side random-policy harness of BASELINE config 2, counterpart of the rollout loop in
 * simulation/avoiding_sim.py:51-66 with agent.predict := U(-0.01, 0.01)^2 (Philox4x32-10, key = seed,
 * counter = (env_offset + env, t)).  policy_begin latches des = TCP xyz for masked envs (avoiding_sim.py:53-54);
 * policy_action advances des_xy and writes actions[n_envs][7]. */
int d3il_policy_begin(d3il_handle h, const uint8_t* env_mask, void* stream);
int d3il_policy_action(d3il_handle h, uint64_t seed, uint64_t env_offset, uint32_t t, double* actions, void* stream);

/* Vectorised-env auto-reset: every environment whose `done` flag is set is reset (as d3il_reset with mask = done; Pushing /
 * Sorting: with the context of that environment's last d3il_reset), the harness re-latches its desired pose (policy_des := TCP,
 * avoiding_sim.py:53-54, pushing_sim.py:69-70, sorting_sim.py:120-121), episode_counts (device i64[2], may be NULL for
 * Pushing / Sorting) += {finished, successful} episodes, and buf.last_reset marks the environments that were reset.
 * Counterpart of starting the next trajectory in the rollout loops (avoiding_sim.py:45-54, pushing_sim.py:43-66). */
int d3il_auto_reset(d3il_handle h, int64_t* episode_counts_device, void* stream);

/* Policy-side helper (batched policy adapters): causal self-attention of the reference's DiffusionGPT
 * (agents/models/beso/agents/diffusion_agents/k_diffusion/score_gpts.py:15-80) for its short token sequences, fused: qkv f32
 * [B * T][3 H D] (query | key | value per token), out f32 [B * T][H D]; softmax over the keys j <= i, scores scaled by
 * 1 / sqrt(D).  Device pointers, T <= 32, D <= 32; needs no handle. */
int d3il_attention_causal_f32(const float* qkv, float* out, int B, int T, int H, int D, void* stream);

/* Policy-side helper: LayerNorm over the last dimension (torch.nn.LayerNorm semantics) for rows of 4 .. 128 floats (a multiple of 4):
 * the transformer of the reference's BESO policy normalises [B * T][120] activations 13 times per denoising call
 * (score_gpts.py:83-115, :353).  x, y f32 [rows][C]; 16-byte aligned device pointers. */
int d3il_layernorm_f32(const float* x, const float* weight, const float* bias, float* y, long rows, int C, float eps, void* stream);
/* Fused transformer MLP of the batched DiffusionGPT (score_gpts.py:83-115: x + fc2(GELU(fc1(h))), h = ln2(x)) on the f32 matrix cores:
 * out[rows][C] = x + b2 + W2 GELU(W1 h + b1).  w_packed = both weight matrices in the per-chunk LDS order of the kernel (d3il_amd/policies.py
 * pack_mlp_weights; H / 16 chunks of 4096 floats).  Built for C = 120, H = 480 (D3IL_EUNSUPPORTED otherwise); all device pointers, 16-byte aligned. */
int d3il_mlp_gelu_residual_f32(const float* h, const float* x, const float* w_packed, const float* b1, const float* b2, float* out, long rows, int C, int H, void* stream);
/* The same with the LayerNorm in front fused in (h = the block's residual stream x itself, ln_weight / ln_bias / ln_eps of ln2; NULL weights = no LayerNorm). */
int d3il_mlp_ln_gelu_residual_f32(const float* h, const float* ln_weight, const float* ln_bias, float ln_eps, const float* x, const float* w_packed, const float* b1, const float* b2,
                                  float* out, long rows, int C, int H, void* stream);
/* out[rows][N] = (LayerNorm)(xin)[rows][120] W^T + bias (+ resid[rows][N]) on the f32 matrix cores: the 120-input linear layers of the DiffusionGPT block
 * (query | key | value as ONE product with N = 360 after ln1; the attention output projection with the residual, N = 120).  w_packed = W [N][120] in the
 * kernel's tile order (d3il_amd/policies.py pack_linear120_weights: ceil(N / 16) tiles of 2048 floats).  ln_weight NULL = no LayerNorm; resid NULL = none. */
int d3il_linear120_f32(const float* xin, const float* ln_weight, const float* ln_bias, float ln_eps, const float* w_packed, const float* bias, const float* resid, float* out,
                       long rows, int N, void* stream);

/* The same two products on the f16 matrix cores with SPLIT operands (csrc/policy_f16x3.h): every f32 operand x = xh + 2^-11 xl as two f16 numbers (22 of the 24
 * significant bits), w x = wh xh + 2^-11 (wh xl + wl xh) in f32 accumulators - three v_mfma_f32_16x16x32_f16 per f32 product instead of sixteen f32-MFMA issue slots.
 * Agreement with an f64 reference is that of an f32 FMA chain (tests/test_policies_f16x3.py); operands saturate at +-65504.  w_packed: f16 halves in the kernels' tile
 * order (d3il_amd/policies.py pack_mlp_weights_f16x3: 16 stages of 2048 x 16 bytes; pack_linear120_weights_f16x3: an even number of tiles of 512 x 16 bytes). */
int d3il_mlp_ln_gelu_residual_f16x3(const float* h, const float* ln_weight, const float* ln_bias, float ln_eps, const float* x, const void* w_packed, const float* b1, const float* b2,
                                    float* out, long rows, int C, int H, void* stream);
int d3il_linear120_f16x3(const float* xin, const float* ln_weight, const float* ln_bias, float ln_eps, const void* w_packed, const float* bias, const float* resid, float* out,
                         long rows, int N, void* stream);
/* The attention half of a DiffusionGPT block (score_gpts.py:35-80 CausalSelfAttention, 103-107 Block.forward first line) in one launch:
 * out = x + proj(causal_attention(LayerNorm(x) Wqkv' + b_qkv)) + b_proj for n_seq sequences of T <= 16 tokens x 120 features, 6 heads; one wave per sequence,
 * q | k | v never leave the CU.  w_packed: the 24 tiles of the packed (query | key | value) weight followed by the 8 tiles of the packed projection weight
 * (policies.pack_linear120_weights_f16x3 of both, concatenated).  out must not alias x. */
int d3il_attn_half_f16x3(const float* x, const float* ln_weight, const float* ln_bias, float ln_eps, const void* w_packed, const float* b_qkv, const float* b_proj, float* out,
                         long n_seq, int T, int n_head, int C, void* stream);

/* The whole sampling chain of the reference's DDPM policy in one launch (agents/models/diffusion/gc_diffusion.py:101-216: epsilon prediction, clipped x0, posterior
 * mean, n_timesteps ancestral steps, final clamp; the denoiser = DiffusionMLPNetwork, diffusion_models.py:20-118 over ResidualMLPNetwork, common/mlp.py:114-182: Linear,
 * n_blocks pre-activation residual blocks with Mish, Linear; window 1), f32 on the matrix cores; rows are independent.  All device pointers, f32:
 *   state [rows][state_dim] (scaled observation), noise [n_timesteps + 1][rows][2] (draw 0 = x_T, draw 1 + k = the k-th step's noise), temb [n_timesteps][8] (the time
 *   embedding of step i, row i), sched [n_timesteps][5] = sqrt(1 / acp), sqrt(1 / acp - 1), posterior mean coefficients 1 and 2, sigma (0 for step 0), bounds = min[2] max[2],
 *   out [rows][2] (scaled action); weights in the kernel's tile order (d3il_amd/policies.py pack_ddpm_weights): w_in [16][64][8], w_blocks [2 n_blocks][16][16][64][4],
 *   w_out [16][64][4]; b_in [256], b_blocks [2 n_blocks][256], b_out [2].  Built for hidden 256, action 2, t_dim 8, state_dim <= 18 (D3IL_EUNSUPPORTED otherwise). */
int d3il_ddpm_mlp_f32(const float* state, const float* noise, const float* temb, const float* w_in, const float* b_in, const float* w_blocks, const float* b_blocks,
                      const float* w_out, const float* b_out, const float* sched, const float* bounds, float* out, long rows, int state_dim, int n_timesteps, int hidden,
                      int n_blocks, void* stream);

/* out[rows][out_dim] = the reference's ResidualMLPNetwork (agents/models/common/mlp.py:114-182: Linear, n_blocks pre-activation residual blocks with Mish, Linear; the
 * network of BC_Agent.predict, bc_agent.py:240-271) of x [rows][in_dim] in one launch on the f32 matrix cores.  Weights in the kernel's tile order
 * (d3il_amd/policies.py pack_resmlp_weights; NT = hidden / 16): w_in [NT][64][8], w_blocks [2 n_blocks][NT][NT][64][4], w_out [NT][64][4]; b_in [hidden],
 * b_blocks [2 n_blocks][hidden], b_out [16] (padded).  Built for hidden 128 / 256, in_dim <= 28, out_dim <= 16 (D3IL_EUNSUPPORTED otherwise). */
int d3il_resmlp_f32(const float* x, const float* w_in, const float* b_in, const float* w_blocks, const float* b_blocks, const float* w_out, const float* b_out, float* out, long rows,
                    int in_dim, int hidden, int n_blocks, int out_dim, void* stream);

/* Per-context episode tally, filled by d3il_auto_reset before it resets: table i64 [n_ctx][D3IL_TALLY_ROW] (caller-owned device
 * memory, caller zeroes it), row ctx_id[env] (device i32[n_envs]; NULL = row 0) += {episodes, successes, successes by mode code}
 * with the mode code = Avoiding: 9-bit mode encoding; Pushing: info['mode'] + 1; Sorting: np.packbits code.  These are the
 * integer tables the metric tails work from (avoiding_sim.py:128-135, pushing_sim.py:140-167, sorting_sim.py:191-208) and the
 * input of the single cross-GPU all-reduce.  table = NULL switches the tally off.
 * Stacking: order code of D3IL_SFLAG_MODE_MASK (< 256) and, because info['success_1'] / ['success_2'] (stacking_sim.py:118-136) also count
 * episodes that did not stack all three boxes, row[2 + D3IL_TALLY_ALL + code] += 1 for EVERY finished episode. */
enum { D3IL_TALLY_ROW = 2 + 512, D3IL_TALLY_ALL = 256 };
int d3il_set_tally(d3il_handle h, const int32_t* ctx_id_device, int n_ctx, int64_t* table_device);

/* Integer metric counts on device: out_counts i64[2 + 512] = {n_done, n_success, histogram of 9-bit mode codes
 * among successful envs}; input to the cross-GPU reduction (one RCCL all-reduce, done by the Python layer)
 * and to success-rate / entropy (avoiding_sim.py:128-135). */
int d3il_count_metrics(d3il_handle h, int64_t* out_counts_device, void* stream);

/* The one exchange step of the path: the int64 metric tables of all GPUs are summed with ONE RCCL all-reduce over xGMI, issued by
 * the library on the caller's stream.  Replaces the shared-memory tensors the reference's worker processes write their results into
 * (avoiding_sim.py:104-118 `successes` / `mode_encoding` with share_memory_(), pushing_sim.py:96-131, sorting_sim.py:144-181,
 * stacking_sim.py:182-216).  RCCL is resolved at run time (an RCCL already in the process - PyTorch's - or librccl.so), so the library
 * does not link it.  Protocol: rank 0 calls d3il_comm_unique_id and hands the 128 bytes to the other ranks by any host channel (bench.py:
 * a torch.distributed broadcast); every rank calls d3il_comm_init(id, rank, world, device_id) - collective -, then
 * d3il_reduce_metrics(h, comm, table, count, stream) in place on device i64[count] (table = NULL: the table of d3il_set_tally), and
 * d3il_comm_destroy.  Integer sums: bit-exact, independent of rank order and of the number of GPUs. */
typedef struct d3il_comm_s* d3il_comm;
typedef struct { char internal[128]; } d3il_rccl_unique_id;   /* = ncclUniqueId */
int d3il_rccl_available(void);                         /* 1 when RCCL could be resolved in this process (no communicator is made) */
int d3il_comm_unique_id(d3il_rccl_unique_id* out);
int d3il_comm_init(const d3il_rccl_unique_id* id, int rank, int world, int device_id, d3il_comm* out);
int d3il_comm_count(d3il_comm comm, int* ranks);       /* ncclCommCount: the number of ranks RCCL itself sees in this communicator */
int d3il_comm_destroy(d3il_comm comm);
int d3il_reduce_metrics(d3il_handle h, d3il_comm comm, int64_t* table_device, size_t count, void* stream);

/* Timing of the last d3il_step kernel launch on its own stream (HIP events recorded around the launch when
 * enabled); used by bench.py for the roofline figure. */
int d3il_set_timing(d3il_handle h, int enabled);
int d3il_last_step_ms(d3il_handle h, float* ms);
/* With timing enabled EVERY d3il_step launch gets its own event pair (a ring of 128; a pair is read when its slot comes round again, so the host never waits
 * for a launch it has just enqueued).  Drains the ring and returns out4 = {sum of the launch durations in ms, min, max, number of launches} since d3il_set_timing(h, 1). */
int d3il_timing_stats(d3il_handle h, double* out4);
/* One host call for one iteration of the rollout loops (fewer trips through the binding when a GPU's environments are stepped as several sub-batches):
 * d3il_step_auto_reset = d3il_step + d3il_auto_reset; d3il_random_rollout_step = d3il_policy_action + d3il_step + d3il_auto_reset (BASELINE config 2). */
int d3il_step_auto_reset(d3il_handle h, const double* actions, int64_t* episode_counts_device, void* stream);
int d3il_random_rollout_step(d3il_handle h, uint64_t seed, uint64_t env_offset, uint32_t t, double* actions, int64_t* episode_counts_device, void* stream);
/* Option "graph_rollout": captures the graphs the next d3il_random_rollout_step calls with these arguments will launch, without launching anything (outside a timed region). */
int d3il_random_rollout_prepare(d3il_handle h, uint64_t seed, uint64_t env_offset, uint32_t t, double* actions, int64_t* episode_counts_device, void* stream);

/* "ik_fast_path" (default 1), "split_waves" (-1 auto, 0, 1), "lanes_per_wave", "lds_pad_bytes";
 * "serve_wave_max_workgroups" (default 256): Avoiding - up to this many workgroups (64 environments each) the split kernel runs with a third wave that executes the
 * rare constraint paths (rod contact, arm joint limits) for the physics wave; above it (more than one workgroup per CU) the two-wave form runs; 0 = always two waves (A/B);
 * "solver_strict" (default 0): 1 = the contact solvers of Pushing / Sorting / Stacking iterate to round-off like the CPU oracle (parity A/B);
 * "stack_reset_coop" (default 1): Stacking env.reset() through the step kernel's wave-cooperative phases, 0 = the one-lane reset kernel (A/B);
 * "graph_rollout" (default 0): Avoiding - d3il_random_rollout_step is captured once per handle (HIP graph: policy kernel with the step counter in device memory,
 * step kernel, mask copy, tally, auto-reset, counter + 1) and a step becomes ONE hipGraphLaunch instead of eight runtime calls; needs a non-null stream
 * (the legacy default stream cannot be captured); any later option / timing / tally change drops the graphs, the next call re-captures.  With timing enabled every
 * eighth step runs uncaptured with the event pair around its step launch (event nodes inside a graph give no timestamps with this runtime): d3il_timing_stats
 * then covers a uniform 1-in-8 sample of the launches;
 * "fuse_rollout_tail" (default 0): Avoiding - d3il_random_rollout_step as TWO launches: the step kernel and one kernel that does everything between two
 * steps (last_reset mask, episode counters and tally of the finished environments, their reset and re-latch, and the policy's action for the NEXT step -
 * which the next call of an uninterrupted sequence finds in `actions`); same results as the five launches and the copy it replaces. */
int d3il_set_option(d3il_handle h, const char* name, int value);
/* Diagnostics builds only (-DD3IL_DEVICE_STATS): per-path lane/wave counters of the step kernel. */
int d3il_debug_stats(uint64_t* out32, int reset);
int d3il_debug_wave_stats(uint64_t* out_nwaves_x10, int nwaves, int reset);
int d3il_debug_wave_counts(uint64_t* out_nwaves_x8, int nwaves, int reset);      /* generic engine: solver event counts per workgroup */
/* Diagnostics: copies `count` doubles of one environment's solver scratch column (contact records of its last physics sub-step)
 * to the host; Pushing / Sorting / Stacking only. */
int d3il_debug_scratch(d3il_handle h, int env, double* out, int count);
const char* d3il_last_error(void);
size_t d3il_blob_sizeof(void);
int d3il_version(void);

#ifdef __cplusplus
}
#endif
#endif
