/*
 * gq_step_kernel.h - the fused batched physics step for gfx950: one environment per 64-lane wavefront,
 * one wavefront per workgroup, per-env working set in LDS, state/obs rows streamed coalesced from/to HBM.
 *
 * Replaces, for a batch of envs, what the reference does per env in QuadrupedEnv.step()
 * (gym_quadruped/quadruped_env.py:270-290): mjData.ctrl = action; mujoco.mj_step; _get_obs; reward;
 * termination checks.  The stages follow MuJoCo's mj_step (restated on the CPU in oracle/gq_oracle.c):
 *
 *   S0 load                state rows -> LDS; actuation and passive damping (mj_fwdActuation, mj_passive)
 *   S1 kinematics          lane = link local transforms, then lanes 0-3 compose the 4 leg chains (mj_kinematics)
 *   S2 spatial inertias    lane = body; composite = plain sums          (mj_comPos, mj_crb)
 *   S3 mass matrix         lane = dof, tree-sparse storage              (mj_crb)
 *   S4 L'DL factorisation  PGS path only: lanes 0-3 = legs, 6x6 base block (mj_factorM), M and M + h*D
 *   S5 bias forces         RNE, lanes 0-3 chains / lane = body          (mj_comVel, mj_rne)
 *   S6 collision           feet spheres + link vertex clouds vs floor   (mj_collision)
 *   S7 constraint rows     lane = row: J row, impedance, R, aref        (mj_makeConstraint, mj_makeImpedance)
 *   S8/S9 Newton (default) primal solve, fused register-resident elimination (gq_newton.h)   (mj_solNewton)
 *   S8/S9 PGS              lane i solves M x = J_i' and owns row i of A = J M^-1 J' + R; sequential rows, residual kept
 *                          in lanes                                     (mj_projectConstraint, mj_solPGS)
 *   S10 integration        semi-implicit Euler with implicit damping    (mj_Euler)
 *   S11 observations       ALL_OBS (+IMU) scalars assembled in LDS, gathered to the requested layout (_get_obs, _check_*)
 *
 * Spatial quantities are expressed in world-aligned axes about O = base-body origin (MuJoCo uses the subtree
 * centre of mass; M, bias forces and Jacobians are independent of that choice) and base x/y are removed from
 * all fp32 arithmetic so that spawning 10 km from the origin (terrain.py:359) costs no precision.
 */
#pragma once
#include <gq_device.h> /* angle brackets: the include path decides (csrc/ for the product, tests/simt_emu/ for the emulator) */
#include "gq_model_dev.h"
#include <cstddef>

namespace gq {

/* StepArgs lives in DEVICE memory (one block per batch, re-uploaded only when a pointer changes): the kernel reads the
 * pointers where it uses them (scalar loads).  Passed by value, the ~60 pointers of a fused step + reset were preloaded
 * into SGPRs at kernel entry and immediately spilled lane-by-lane into VGPRs (v_writelane / v_readlane: ~10 % of the
 * kernel's VALU issue slots).  What changes from call to call travels by value in StepCall. */
/* Field order (round 5): the sixteen pointers a wave's PROLOGUE reads (load_rows) are the first 128 bytes - two s_load_dwordx16 - and the
 * ones its EPILOGUE stores through follow in one block (from `qacc` on); the block is filled by name (gq_api.hip). */
struct StepArgs {
  const GqDevModel* model;
  const GqDevBatch* batch;
  const float* vx; const float* vy; const float* vz; /* cloud vertices SoA */
  double* qpos; float* qvel; float* warm; const float* applied; float* time; const float* friction;
  const float* cmd;
  uint8_t* pending;       /* library scratch [N]: env terminated and waits for its next-step auto-reset (may be NULL) */
  uint8_t* load_hint;     /* library scratch [N]: Newton iterations of the env's previous step, capped (may be NULL) */
  int32_t* step_num;
  const uint8_t* lift_pending; /* library scratch [N] written by reset_kernel (see ResetArgs), read by first-pass steps; may be NULL */
  float timestep; int32_t nlg, nfl, pad_; /* copies of the model's scalars every stage reads: they arrive with the pointers, not behind the model pointer */
  /* ---- epilogue */
  float* qacc;
  float* obs; float* reward; uint8_t* terminated; uint8_t* truncated; uint8_t* invalid_contact;
  int32_t* step_prev;     /* [N] step counter before this step's increment (info['step_num']), may be NULL */
  float* imu_bias;        /* [N][6] accelerometer / gyro bias random walks (in/out), NULL = no IMU */
  const int32_t* episode_ro; /* [N] episode counters (RNG counter word), may be NULL */
  float* friction_next;   /* library scratch [N]: friction drawn at reset, committed after the reset's own step (:403-404) */
  uint8_t* lift_failed;   /* [N] out: the reset RuntimeError condition (:387-388), written by the step that performs a lift; may be NULL */
  int32_t* contacts_dropped; /* [N] contacts of this step's narrow phase that did not fit the 12-contact / 63-row capacity, may be NULL */
  int32_t* h9;            /* [N][6] resampling counters {after_vel, before_vel, n_vel, after_dist, before_dist, n_dist}, may be NULL */
  float* ext_dist;        /* [N][6] current disturbance wrench, may be NULL */
  float* dyn;             /* [N][GQ_DYN_STRIDE] dynamics rows (gq_batch_set_outputs), may be NULL */
  float* contacts;        /* [N][GQ_CON_STRIDE] contact rows, may be NULL */
  float* heightmap;       /* [N][rows * cols][3] hit points of the HeightMap that follows the base (gq_batch_set_heightmap), may be NULL */
  int32_t n_envs;
};
/* layout of the convex pair exchange (gq_exchange.h), int32 words: [slots] state words, then [slots][GQ_XQ_ITEM] items */
#define GQ_XQ_ITEM 64
#define GQ_XQ_MARGIN 40
#define GQ_XQ_RES 48 /* hit, dist, normal (3), point (3) */
struct StepCall {
  const float* ctrl;    /* [N][nu] or NULL (zero control) */
  const uint8_t* mask;  /* [N] or NULL */
  float* debug;         /* debug record block or NULL */
  int32_t auto_reset;   /* 0 off, 1 same-step (second pass in this launch), 2 next-step (pending flag, one pass per launch) */
  int32_t first_pass;   /* 0: user step; 1: the reset's own step (gq_reset) */
  int32_t env0;         /* first env of this launch (gq_step_range); env = env0 + blockIdx.x */
  int32_t forward;      /* gq_forward (instrumented variant only): 1 = mj_step1 (return after the constraint rows), 2 = mj_forward (return
                         * after the accelerations); nothing but qacc and the inspection record is written */
  /* persistent open-loop rollout (gq_rollout with shards = 0): every wavefront plays n_steps steps of ITS env back to back - no
   * launch boundary, hence no wave ever waits for another env's stragglers; step k takes its controls from ctrl + k * ctrl_stride
   * and, when obs_seq is given, also writes its observation row to obs_seq + (k * n_envs + env) * obs_dim */
  int32_t n_steps;      /* 0 / 1: a single step */
  int32_t ctrl_stride;  /* floats between the control rows of consecutive steps */
  float* obs_seq;
  int32_t count;        /* envs of this launch: wavefronts past env0 + count (the last workgroup of a multi-wave launch) return at once */
  /* closed loop INSIDE the persistent rollout (gq_rollout_closed, inline mode): the wavefront derives the action of its env's next
   * step from the observation row it has just written (PolicyPdDev below, device memory); controls are then not read from ctrl */
  const struct PolicyPdDev* policy;
  float* act_seq;       /* [n_steps][N][12] every action taken, or NULL */
  int32_t stop_stage;   /* profiling aid (env GQ_STOP_STAGE, tools/stage_insts.sh): return after stage marker i; 0 = run everything */
};

/* ---- closed-loop persistent rollout (gq_rollout_closed): env-steps are TASKS.  A policy - a kernel of its own on a second stream,
 * or anything else that follows the protocol - turns the observation an env published after step k - 1 into the action of step k,
 * writes it into the env's action mailbox and pushes the env onto a ready queue; the wavefronts of ONE persistent step launch pop
 * ready envs, play one step each (QuadrupedEnv.step() semantics per env, next-step auto-reset included), publish the observation
 * row and the env's step count, and pop again.  No env waits for another env's Newton tail or for a launch boundary, any number of
 * envs works with any number of resident wavefronts, and nothing can deadlock: every wait has a deadline that raises the abort word.
 * One queue per XCD: an env is always stepped by wavefronts of the same XCD, so its state rows stay coherent in that XCD's L2 and
 * only the mailbox words (actions, observation rows, sequence numbers, queue slots) travel with device-coherent accesses. */
#define GQ_MB_QSTRIDE 32  /* int32 words between two queue counters: a 128-byte line each */
struct MailboxDev {
  float* act;             /* [N][12] action mailbox: written by the policy, read by the wavefront that plays the env's next step */
  int32_t* steps_done;    /* [N] steps the env has completed in this rollout: published AFTER its observation row */
  int32_t* issued;        /* [N] actions the built-in policy has issued per env (its private scratch) */
  int32_t* q_items;       /* [nq][qcap] ring of env + 1 (0: empty slot); qcap is a power of two >= N, an env is queued at most once */
  int32_t* q_ctr;         /* [nq][3][GQ_MB_QSTRIDE]: pop tickets, push tickets, policy wavefronts registered on the queue's XCD */
  int32_t* status;        /* [8] word 0: abort code (0 running / fine, 1 a step wavefront waited past the deadline, 2 the policy did),
                           * word 1: the ticket / policy lane that gave up, word 2: env-steps played */
  int32_t* alive;         /* pinned host word: policy workgroups that are resident (the host waits for it before the step launch) */
  float* act_seq;         /* [K][N][12] every action taken, or NULL */
  float* obs_seq;         /* [K][N][obs_dim] every observation row, or NULL */
  int32_t n_envs, n_steps, qcap, nq;
  int32_t xcc_queue[16];  /* HW_REG_XCC_ID -> queue */
  int64_t timeout_ticks;  /* deadline of every wait, 100 MHz ticks */
  int32_t flags;          /* experiment switches (env GQ_MB_FLAGS): 1 policy: device-scope release fence before the item store; 2 step: device-scope
                           * acquire fence after the item; 4 step: release fence before the count; 8 policy: acquire fence after the count */
};
/* built-in policy: joint-space PD towards a posture, torque_j = kp_j (q_des_j - q_j) - kd_j qd_j (rounded after every operation,
 * like the elementwise torch expression); col_*: columns of the joint angles / velocities in the observation row */
struct PolicyPdDev {
  float kp[12], kd[12], qdes[12]; int32_t col_q[12], col_qd[12];
  /* Gaussian exploration noise added to the law: sigma * N(0, 1) per joint and step, Philox4x32-10 block (joint, step0 + k, global env
   * id, 0x9011), key = seed (0 sigma: none) */
  float sigma; uint32_t seed_lo, seed_hi; int32_t step0, env_id_offset;
};

/* wave-uniform model scalars of S5 - S9 (and the cloud pointers): one batch of scalar loads in front of S5, pinned */
struct StepConsts {
  int foot_leg[4], iterations, nsp, self_cut;
  float floor_mu, impratio_rs /* 1 / sqrt(impratio) */, gravity_z, nw_scale /* 1 / (meaninertia nv) */, tolerance, noise_floor, self_margin;
  const GQ_MODEL float* vx; const GQ_MODEL float* vy; const GQ_MODEL float* vz;
};

/* canonical ALL_OBS scalar offsets (order of QuadrupedEnv.ALL_OBS, quadruped_env.py:35-66,81) */
enum {
  OB_BASE_POS = 0, OB_LIN_VEL = 3, OB_LIN_VEL_ERR = 6, OB_LIN_ACC = 9, OB_ANG_VEL = 12, OB_ANG_VEL_ERR = 15,
  OB_EULER = 18, OB_QUAT = 21, OB_SO3 = 25, OB_GRAV_B = 34, OB_LIN_VEL_B = 37, OB_LIN_VEL_ERR_B = 40,
  OB_LIN_ACC_B = 43, OB_ANG_VEL_B = 46, OB_ANG_VEL_ERR_B = 49, OB_QPOS = 52, OB_QVEL = 71, OB_TAU = 89,
  OB_QPOS_JS = 101, OB_QVEL_JS = 113, OB_KE = 125, OB_WORK = 126, OB_FEET_POS = 127, OB_FEET_POS_B = 139,
  OB_FEET_VEL = 151, OB_FEET_VEL_REL = 163, OB_FEET_VEL_B = 175, OB_FEET_VEL_REL_B = 187, OB_CONTACT_STATE = 199,
  OB_CONTACT_F = 203, OB_CONTACT_F_B = 215,
  OB_IMU_ACC = 227, OB_IMU_ACC_NOISE = 230, OB_IMU_ACC_BIAS = 233, OB_IMU_GYRO = 236, OB_IMU_GYRO_NOISE = 239,
  OB_IMU_GYRO_BIAS = 242
};

/* flat layout of one tree-sparse L'DL factor in LDS */
#define GQ_F_LC(k, j) ((k) * 8 + (j))          /* leg dof 6+k, column j: 0-5 base, 6 hip, 7 thigh */
#define GQ_F_LB(i, j) (96 + (i) * 6 + (j))     /* base block, lower */
#define GQ_F_DINV(k) (132 + (k))
#define GQ_FACTOR_SIZE 150

enum { ROW_NONE = 0, ROW_FRICTION = 1, ROW_LIMIT = 2, ROW_CONTACT1 = 3, ROW_PYRAMID = 4, ROW_ELLIPTIC = 5 };

/* ------------------------------------------------------------------ per-wave LDS working set (10 240 B, all of it used: 16 waves per CU fit the 160 KB)
 * `u` overlays three regions with disjoint lifetimes: the spatial-dynamics scratch (S1-S5), the half-batch of
 * B = M^-1 J' rows while the dual operator is built (S8), and the observation row (S11). */
struct WaveDyn {
  float anchor[GQ_NJ][3], axis[GQ_NJ][3];
  float cinert[GQ_NB][10];
  union { float crb[GQ_NB][10]; float acc[8][21]; };    /* acc: per-leg Schur updates of the base block (S4, crb is dead) */
  union {
    struct { float cvel[GQ_NB][6], cacc[GQ_NB][6]; };
    float fkloc[GQ_NJ][13];   /* S1 only: per-link local transform (quat 4, offset 3, anchor 3, axis 3) */
  };
  float cfrc[GQ_NB][6];
};
/* Development aid (tools/dev_build.sh ... -DGQ_TICKSET=<n>, instrumented variant only): the 13 stage stamps of the debug record are moved to
 * the sub-stage points GQ_SUB(W, n, k) of set n - a finer cut of one half of the step for tools/perf_probe.py substages.  0: not compiled in. */
#ifndef GQ_TICKSET
#define GQ_TICKSET 0
#endif
#if GQ_TICKSET
#define GQ_SUB(W, SET, K) do { if constexpr (GQ_TICKSET == (SET)) { if ((W).tk_T && lane_id() == 0) { constexpr int ord_[13] = {1, 2, 3, 4, 5, 14, 6, 7, 8, 9, 10, 11, 12}; \
    (W).tk_T[ord_[K]] = (float)((long long)__builtin_readcyclecounter() - (W).tk_t0); } } } while (0)
#else
#define GQ_SUB(W, SET, K) do { } while (0)
#endif
struct WaveMem {
#if GQ_TICKSET
  float* tk_T; long long tk_t0;
#endif
  double bxy[2];               /* base x, y of this forward pass (f64, never enters fp32 arithmetic) */
  float mu_env; int32_t step_old; /* the env's friction override (-1: none) and its step counter before this step */
  float qj[12], qb[4], basez, qvel[18], ctrl[12], warm[18], cmd[4];
  float xpos[GQ_NB][3];
  union { float xmat[GQ_NB][9]; float acc2[5][21]; };   /* acc2: Newton factor/solve exchange (xmat is dead after S6) */
  float cdof[GQ_NVD][6];
  /* joint-space inertia, tree-sparse: leg dof 6+j keeps [b0..b5, hip, thigh, calf] of its own leg (lower part incl.
   * the diagonal), the base block is a full symmetric 6x6 */
  float Mc[GQ_NJ][9], Mb[6][6];
  /* L'DL factors, tree-sparse storage: leg dof k >= 6 keeps [b0..b5, hip, thigh] (8 floats), base rows 6x6 lower.
   * [0]: M, [1]: M + h*diag(damping) */
  float F[2][GQ_FACTOR_SIZE];  /* per factor: Lc[12][8] | Lb[6][6] | Dinv[18], see the GQ_F_* offsets */
  float bias[18], act[18], smooth[18], qacc_smooth[18], qfrc_c[18], qacc[18], qacc_int[18];
  /* contacts */
  int32_t ncon, nefc, nlim, invalid;
  int32_t nself;               /* number of robot-robot contacts in the list (S6, BOXES variants) */
  int32_t foot_touch;          /* bit k: foot k's body touches a world geom (also when the contact fell to the row budget) */
  int32_t ndrop;               /* contacts found by the narrow phase that the 12-contact / 63-row capacity cut (S6; info["contacts_dropped"]) */
  int32_t con_geom[GQ_MAXCON], con_body[GQ_MAXCON], con_dim[GQ_MAXCON], con_row[GQ_MAXCON];
  float con_dist[GQ_MAXCON], con_pos[GQ_MAXCON][3], con_mu[GQ_MAXCON], con_inc[GQ_MAXCON];
  float con_solref[GQ_MAXCON][2], con_solimp[GQ_MAXCON][5];
  float con_t1[GQ_MAXCON][2];  /* floor contacts: first tangent (cos, sin, 0) of the contact frame - (0, 1) = mju_makeFrame's default
                                * y axis; mjraw_PlaneCapsule aligns it with the capsule axis */
  float foot_world[4][3];
  union {                                               /* collision / limit scratch (S6-S7)  |  Newton scratch (S9) */
    struct { int32_t lim_jnt[GQ_NJ]; float lim_side[GQ_NJ], lim_dist[GQ_NJ]; float lg_dist[GQ_MAXLG], lg_pt[GQ_MAXLG][3]; } c;
    struct { float Hc[GQ_NJ][9], Hb[6][6], nw[2][GQ_NVD]; } n;
  } u2;
  float force[64];
  union {
    WaveDyn dyn;
    float B[64][GQ_NVD];                                /* rows of M^-1 J' while A is built (S8) */
    float obs[256];
  } u;
};
static_assert(sizeof(WaveMem) <= 10240 || GQ_TICKSET != 0, "16 one-wave workgroups per CU need <= 10 240 B of LDS each: WaveMem is full, a new field must reuse a dead one");
/* S0 .. S6: whether the reset's lift loop is due in this step's S6, as an int in force[1] (free until the solver, like the clock in force[0];
 * carried in a register from the prologue across the re-spawn branch it was spilled to scratch memory, and WaveMem has no byte to spare) */
#define GQ_LIFT_DUE(W) (reinterpret_cast<int32_t*>(&(W).force[1])[0])

/* ------------------------------------------------------------------ small math */
struct V3 { float x, y, z; };
__device__ __forceinline__ V3 v3(float x, float y, float z) { V3 r = {x, y, z}; return r; }
template <class P> __device__ __forceinline__ V3 ld3(P p) { return v3(p[0], p[1], p[2]); } /* any address space */
__device__ __forceinline__ void st3(float* p, V3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
template <class P> __device__ __forceinline__ V3 matvec(P m, V3 v) {
  return v3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[3] * v.x + m[4] * v.y + m[5] * v.z, m[6] * v.x + m[7] * v.y + m[8] * v.z);
}
template <class P> __device__ __forceinline__ V3 matTvec(P m, V3 v) {
  return v3(m[0] * v.x + m[3] * v.y + m[6] * v.z, m[1] * v.x + m[4] * v.y + m[7] * v.z, m[2] * v.x + m[5] * v.y + m[8] * v.z);
}
struct Q4 { float w, x, y, z; };
__device__ __forceinline__ Q4 qmul(Q4 a, Q4 b) {
  Q4 r = {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
          a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
  return r;
}
/* atan2 for finite arguments: odd minimax polynomial of degree 15 on [0, 1] (|error| < 1.2e-7 rad) after the usual
 * octant reduction; the library routine's handling of infinities, NaNs and denormals is not needed for rotation-matrix
 * entries, and it is three times the instructions */
__device__ __forceinline__ float atan2_fast(float y, float x) {
  const float ax = fabsf(x), ay = fabsf(y), mx = fmaxf(ax, ay), mn = fminf(ax, ay);
  const float a = mx > 0.0f ? mn * fast_rcp(mx) : 0.0f, s = a * a;
  float p = -0.0040543945506215096f;
  p = fmaf(p, s, 0.021862290799617767f); p = fmaf(p, s, -0.055911291390657425f); p = fmaf(p, s, 0.09642115235328674f);
  p = fmaf(p, s, -0.13908594846725464f); p = fmaf(p, s, 0.1994655728340149f); p = fmaf(p, s, -0.33329859375953674f);
  p = fmaf(p, s, 0.9999993443489075f);
  float r = p * a;
  if (ay > ax) r = 1.57079632679489662f - r;
  if (x < 0.0f) r = 3.14159265358979324f - r;
  return y < 0.0f ? -r : r;
}

/* sin and cos of a moderate angle (joint half-angles, integration increments, yaw: |x| well below 1e3 rad): two-term
 * Cody-Waite reduction to [-pi/4, pi/4] and the classic single-precision minimax kernels, ~30 VALU for both values
 * and 1-2 ulp - the libm routines carry a large-argument reduction that is never needed here */
__device__ __forceinline__ void sincos_small(float x, float& s, float& c) {
  const float kf = rintf(x * 0.636619772367581343f);
  float r = fmaf(-kf, 1.5707962512969970703125f, x);   /* pi/2 split: high 24 bits + remainder */
  r = fmaf(-kf, 7.54978995489188e-08f, r);
  const float z = r * r;
  const float sp = r + r * z * (-0.166666666416265235595f + z * (0.0083333293858894631756f + z * (-0.000198393348360966317347f + z * 0.0000027183114939898219064f)));
  const float cp = 1.0f + z * (-0.499999997251031003120f + z * (0.0416666233237390631894f + z * (-0.00138867637746099294692f + z * 0.0000243904487962774090654f)));
  const int q = (int)kf & 3;
  const float ss = (q & 1) ? cp : sp, cc = (q & 1) ? sp : cp;
  s = (q & 2) ? -ss : ss;
  c = ((q + 1) & 2) ? -cc : cc;
}
__device__ __forceinline__ Q4 qnormalize(Q4 q) {
  float n2 = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
  if (n2 < 1e-30f) { Q4 i = {1, 0, 0, 0}; return i; }
  float s = fast_rsqrt(n2);
  Q4 r = {q.w * s, q.x * s, q.y * s, q.z * s};
  return r;
}
__device__ __forceinline__ void q2mat(float* m, Q4 q) {
  float w = q.w, x = q.x, y = q.y, z = q.z;
  m[0] = w * w + x * x - y * y - z * z; m[4] = w * w - x * x + y * y - z * z; m[8] = w * w - x * x - y * y + z * z;
  m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y); m[3] = 2 * (x * y + w * z);
  m[5] = 2 * (y * z - w * x); m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x);
}
/* spatial vectors [ang(3); lin(3)] */
__device__ __forceinline__ void cross_motion(float* r, const float* v, const float* s) {
  V3 w = ld3(v), l = ld3(v + 3), sa = ld3(s), sl = ld3(s + 3);
  st3(r, cross(w, sa));
  st3(r + 3, cross(w, sl) + cross(l, sa));
}
__device__ __forceinline__ void cross_force(float* r, const float* v, const float* f) {
  V3 w = ld3(v), l = ld3(v + 3), fa = ld3(f), fl = ld3(f + 3);
  st3(r, cross(w, fa) + cross(l, fl));
  st3(r + 3, cross(w, fl));
}
__device__ __forceinline__ void mul_inert(float* r, const float* i, const float* v) {
  r[0] = i[0] * v[0] + i[3] * v[1] + i[4] * v[2] - i[8] * v[4] + i[7] * v[5];
  r[1] = i[3] * v[0] + i[1] * v[1] + i[5] * v[2] + i[8] * v[3] - i[6] * v[5];
  r[2] = i[4] * v[0] + i[5] * v[1] + i[2] * v[2] - i[7] * v[3] + i[6] * v[4];
  r[3] = i[8] * v[1] - i[7] * v[2] + i[9] * v[3];
  r[4] = i[6] * v[2] - i[8] * v[0] + i[9] * v[4];
  r[5] = i[7] * v[0] - i[6] * v[1] + i[9] * v[5];
}

__device__ __forceinline__ uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32); }
/* Philox4x32-10 (Salmon et al. 2011); returns component `which` of the output block */
__device__ inline uint32_t philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, int which) {
#pragma unroll
  for (int r = 0; r < 10; r++) {
    uint32_t hi0 = mulhi32(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    uint32_t hi1 = mulhi32(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return which == 0 ? c0 : (which == 1 ? c1 : (which == 2 ? c2 : c3));
}

/* standard normal from one Philox block (Box-Muller on words 0,1): u1 in (0,1], u2 in [0,1) */
__device__ __forceinline__ float philox_normal(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
  const uint32_t x0 = philox4x32(c0, c1, c2, c3, k0, k1, 0), x1 = philox4x32(c0, c1, c2, c3, k0, k1, 1);
  const float u1 = ((float)(x0 >> 8) + 0.5f) * (1.0f / 16777216.0f), u2 = (float)(x1 >> 8) * (1.0f / 16777216.0f);
  return sqrtf(-2.0f * logf(u1)) * fast_cos_turns(u2); /* cos(2 pi u2): v_cos_f32 takes its argument in turns (libm's cosf brought 110
                                                         * instructions of Payne-Hanek range reduction into every kernel variant) */
}

/* dof tree of the fixed topology: parent of dof d */
__device__ __forceinline__ int dof_parent(int d) { return d < 6 ? d - 1 : ((d - 6) % 3 == 0 ? 5 : d - 1); }
__device__ __forceinline__ int dof_body(int d) { return d < 6 ? 0 : d - 5; }

/* solimp impedance (mj_makeImpedance::getimpedance) */
__device__ __forceinline__ float impedance(const float* solimp, float pos, float margin) {
  float dmin = fminf(fmaxf(solimp[0], 0.0001f), 0.9999f), dmax = fminf(fmaxf(solimp[1], 0.0001f), 0.9999f);
  float width = fmaxf(1e-15f, solimp[2]), mid = fminf(fmaxf(solimp[3], 0.0001f), 0.9999f), power = fmaxf(1.0f, solimp[4]);
  if (dmin == dmax || width <= 1e-15f) return 0.5f * (dmin + dmax);
  float x = fdiv(fabsf(pos - margin), width);
  if (x >= 1.0f) return dmax;
  if (x <= 0.0f) return dmin;
  float y;
  if (power == 1.0f) y = x;
  else if (power == 2.0f) y = x <= mid ? fdiv(x * x, mid) : 1.0f - fdiv((1.0f - x) * (1.0f - x), 1.0f - mid); /* MuJoCo's default */
  else if (x <= mid) y = fast_pow_ratio(x, power, mid, power - 1.0f);
  else y = 1.0f - fast_pow_ratio(1.0f - x, power, 1.0f - mid, power - 1.0f);
  return dmin + y * (dmax - dmin);
}

/* L'DL of M (factor 0) and of M + h*diag(damping) (factor 1, the Euler system) with the robot's dof-tree sparsity
 * (mj_factorI), both at once: lanes 0-3 eliminate the three dofs of their leg for factor 0 in registers, lanes 4-7 do
 * the same for factor 1, each emitting its Schur contribution to the 6x6 base block; lanes 0 and 4 then factor the
 * two base blocks.  Reciprocals use v_rcp_f32 (1 ulp). */
__device__ inline void factor_tree_both(WaveMem& W, const GQ_MODEL float* damping, const float h) {
  const int lane = lane_id();
  const int which = (lane >> 2) & 1, leg = lane & 3;
  const float hscale = which ? h : 0.0f;
  float* F = W.F[which];
  float(*acc)[21] = W.u.dyn.acc;
  if (lane < 8) {
    const int hh = 6 + 3 * leg, t = hh + 1, c = hh + 2;
    /* rows over columns [b0..b5, h, t, c] */
    float rc[9], rt[8], rh[7], bb[21];
#pragma unroll
    for (int j = 0; j < 6; j++) { rc[j] = W.Mc[c - 6][j]; rt[j] = W.Mc[t - 6][j]; rh[j] = W.Mc[hh - 6][j]; }
    rc[6] = W.Mc[c - 6][6]; rc[7] = W.Mc[c - 6][7]; rc[8] = W.Mc[c - 6][8] + hscale * damping[c];
    rt[6] = W.Mc[t - 6][6]; rt[7] = W.Mc[t - 6][7] + hscale * damping[t];
    rh[6] = W.Mc[hh - 6][6] + hscale * damping[hh];
#pragma unroll
    for (int q = 0; q < 21; q++) bb[q] = 0.0f;
    const float ic = fast_rcp(rc[8]);
    { /* eliminate calf: ancestors t(7), h(6), b5..b0 */
      float tmp = rc[7] * ic;
#pragma unroll
      for (int j = 0; j <= 7; j++) rt[j] -= rc[j] * tmp;
      rc[7] = tmp;
      tmp = rc[6] * ic;
#pragma unroll
      for (int j = 0; j <= 6; j++) rh[j] -= rc[j] * tmp;
      rc[6] = tmp;
#pragma unroll
      for (int i = 5; i >= 0; i--) {
        tmp = rc[i] * ic;
#pragma unroll
        for (int j = 0; j <= i; j++) bb[i * (i + 1) / 2 + j] -= rc[j] * tmp;
        rc[i] = tmp;
      }
    }
    const float it = fast_rcp(rt[7]);
    { /* thigh: ancestors h(6), b5..b0 */
      float tmp = rt[6] * it;
#pragma unroll
      for (int j = 0; j <= 6; j++) rh[j] -= rt[j] * tmp;
      rt[6] = tmp;
#pragma unroll
      for (int i = 5; i >= 0; i--) {
        tmp = rt[i] * it;
#pragma unroll
        for (int j = 0; j <= i; j++) bb[i * (i + 1) / 2 + j] -= rt[j] * tmp;
        rt[i] = tmp;
      }
    }
    const float ih = fast_rcp(rh[6]);
    { /* hip: ancestors b5..b0 */
#pragma unroll
      for (int i = 5; i >= 0; i--) {
        const float tmp = rh[i] * ih;
#pragma unroll
        for (int j = 0; j <= i; j++) bb[i * (i + 1) / 2 + j] -= rh[j] * tmp;
        rh[i] = tmp;
      }
    }
#pragma unroll
    for (int j = 0; j < 6; j++) { F[GQ_F_LC(c - 6, j)] = rc[j]; F[GQ_F_LC(t - 6, j)] = rt[j]; F[GQ_F_LC(hh - 6, j)] = rh[j]; }
    F[GQ_F_LC(c - 6, 6)] = rc[6]; F[GQ_F_LC(c - 6, 7)] = rc[7]; F[GQ_F_LC(t - 6, 6)] = rt[6];
    F[GQ_F_DINV(c)] = ic; F[GQ_F_DINV(t)] = it; F[GQ_F_DINV(hh)] = ih;
#pragma unroll
    for (int q = 0; q < 21; q++) acc[lane][q] = bb[q];
  }
  wave_barrier();
  if (lane == 0 || lane == 4) {
    float b[6][6];
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
      for (int j = 0; j <= i; j++) {
        const int q = i * (i + 1) / 2 + j;
        b[i][j] = W.Mb[i][j] + (i == j ? hscale * damping[i] : 0.0f) + acc[lane][q] + acc[lane + 1][q] + acc[lane + 2][q] + acc[lane + 3][q];
      }
#pragma unroll
    for (int k = 5; k >= 0; k--) {
      const float inv = fast_rcp(b[k][k]);
#pragma unroll
      for (int i = k - 1; i >= 0; i--) {
        const float tmp = b[k][i] * inv;
#pragma unroll
        for (int j = 0; j <= i; j++) b[i][j] -= b[k][j] * tmp;
        b[k][i] = tmp;
      }
      F[GQ_F_DINV(k)] = inv;
    }
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
      for (int j = 0; j < i; j++) F[GQ_F_LB(i, j)] = b[i][j];
  }
  wave_barrier();
}

/* x <- (L' D L)^-1 x on 18 registers per lane (mj_solveLD).  The 150 factor words are fetched from LDS once,
 * coalesced (3 words per lane), and every coefficient is then broadcast with v_readlane_b32 into the scalar operand
 * of the FMA - no LDS latency inside the dependent chains.  All lanes run the solve, each on its own right-hand side. */
struct FactorRegs { float f0, f1, f2; };
__device__ __forceinline__ FactorRegs load_factor(const WaveMem& W, int which) {
  const int lane = lane_id();
  FactorRegs r;
  r.f0 = W.F[which][lane];
  r.f1 = W.F[which][64 + lane];
  r.f2 = lane < GQ_FACTOR_SIZE - 128 ? W.F[which][128 + lane] : 0.0f;
  return r;
}
template <int E>
__device__ __forceinline__ float fcoef(const FactorRegs& r) {
  return readlane<(E & 63)>(E < 64 ? r.f0 : (E < 128 ? r.f1 : r.f2));
}
#define LCF(k, j) fcoef<GQ_F_LC(k, j)>(fr)
#define LBF(i, j) fcoef<GQ_F_LB(i, j)>(fr)
#define DIF(k) fcoef<GQ_F_DINV(k)>(fr)

template <int LEG>
__device__ __forceinline__ void solve_back_leg(const FactorRegs& fr, float* x) {
  constexpr int h = 6 + 3 * LEG, t = h + 1, c = h + 2;
  x[t] -= LCF(c - 6, 7) * x[c]; x[h] -= LCF(c - 6, 6) * x[c];
  x[0] -= LCF(c - 6, 0) * x[c]; x[1] -= LCF(c - 6, 1) * x[c]; x[2] -= LCF(c - 6, 2) * x[c];
  x[3] -= LCF(c - 6, 3) * x[c]; x[4] -= LCF(c - 6, 4) * x[c]; x[5] -= LCF(c - 6, 5) * x[c];
  x[h] -= LCF(t - 6, 6) * x[t];
  x[0] -= LCF(t - 6, 0) * x[t]; x[1] -= LCF(t - 6, 1) * x[t]; x[2] -= LCF(t - 6, 2) * x[t];
  x[3] -= LCF(t - 6, 3) * x[t]; x[4] -= LCF(t - 6, 4) * x[t]; x[5] -= LCF(t - 6, 5) * x[t];
  x[0] -= LCF(h - 6, 0) * x[h]; x[1] -= LCF(h - 6, 1) * x[h]; x[2] -= LCF(h - 6, 2) * x[h];
  x[3] -= LCF(h - 6, 3) * x[h]; x[4] -= LCF(h - 6, 4) * x[h]; x[5] -= LCF(h - 6, 5) * x[h];
}
template <int LEG>
__device__ __forceinline__ void solve_fwd_leg(const FactorRegs& fr, float* x) {
  constexpr int h = 6 + 3 * LEG, t = h + 1, c = h + 2;
  x[h] -= LCF(h - 6, 0) * x[0] + LCF(h - 6, 1) * x[1] + LCF(h - 6, 2) * x[2] + LCF(h - 6, 3) * x[3] + LCF(h - 6, 4) * x[4] + LCF(h - 6, 5) * x[5];
  x[t] -= LCF(t - 6, 0) * x[0] + LCF(t - 6, 1) * x[1] + LCF(t - 6, 2) * x[2] + LCF(t - 6, 3) * x[3] + LCF(t - 6, 4) * x[4] + LCF(t - 6, 5) * x[5] + LCF(t - 6, 6) * x[h];
  x[c] -= LCF(c - 6, 0) * x[0] + LCF(c - 6, 1) * x[1] + LCF(c - 6, 2) * x[2] + LCF(c - 6, 3) * x[3] + LCF(c - 6, 4) * x[4] + LCF(c - 6, 5) * x[5] + LCF(c - 6, 6) * x[h] + LCF(c - 6, 7) * x[t];
}

__device__ __forceinline__ void solve_tree(const WaveMem& W, int which, float* x) {
  const FactorRegs fr = load_factor(W, which);
  solve_back_leg<3>(fr, x); sched_fence(); solve_back_leg<2>(fr, x); sched_fence();
  solve_back_leg<1>(fr, x); sched_fence(); solve_back_leg<0>(fr, x); sched_fence();
  x[0] -= LBF(5, 0) * x[5]; x[1] -= LBF(5, 1) * x[5]; x[2] -= LBF(5, 2) * x[5]; x[3] -= LBF(5, 3) * x[5]; x[4] -= LBF(5, 4) * x[5];
  x[0] -= LBF(4, 0) * x[4]; x[1] -= LBF(4, 1) * x[4]; x[2] -= LBF(4, 2) * x[4]; x[3] -= LBF(4, 3) * x[4];
  x[0] -= LBF(3, 0) * x[3]; x[1] -= LBF(3, 1) * x[3]; x[2] -= LBF(3, 2) * x[3];
  x[0] -= LBF(2, 0) * x[2]; x[1] -= LBF(2, 1) * x[2];
  x[0] -= LBF(1, 0) * x[1];
  sched_fence();
  x[0] *= DIF(0); x[1] *= DIF(1); x[2] *= DIF(2); x[3] *= DIF(3); x[4] *= DIF(4); x[5] *= DIF(5);
  x[6] *= DIF(6); x[7] *= DIF(7); x[8] *= DIF(8); x[9] *= DIF(9); x[10] *= DIF(10); x[11] *= DIF(11);
  x[12] *= DIF(12); x[13] *= DIF(13); x[14] *= DIF(14); x[15] *= DIF(15); x[16] *= DIF(16); x[17] *= DIF(17);
  sched_fence();
  x[1] -= LBF(1, 0) * x[0];
  x[2] -= LBF(2, 0) * x[0] + LBF(2, 1) * x[1];
  x[3] -= LBF(3, 0) * x[0] + LBF(3, 1) * x[1] + LBF(3, 2) * x[2];
  x[4] -= LBF(4, 0) * x[0] + LBF(4, 1) * x[1] + LBF(4, 2) * x[2] + LBF(4, 3) * x[3];
  x[5] -= LBF(5, 0) * x[0] + LBF(5, 1) * x[1] + LBF(5, 2) * x[2] + LBF(5, 3) * x[3] + LBF(5, 4) * x[4];
  sched_fence();
  solve_fwd_leg<0>(fr, x); sched_fence(); solve_fwd_leg<1>(fr, x); sched_fence();
  solve_fwd_leg<2>(fr, x); sched_fence(); solve_fwd_leg<3>(fr, x); sched_fence();
}
#undef LCF
#undef LBF
#undef DIF

}  // namespace gq
