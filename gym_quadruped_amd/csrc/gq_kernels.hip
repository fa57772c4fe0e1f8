/*
 * gq_kernels.hip - gfx950 kernels of libgq: one environment per 64-lane wavefront, one wavefront per workgroup
 * (grid = n_envs).  With 4096 envs and <= 128 VGPRs / <= 10 KB LDS per wave the whole batch is resident at once
 * (4 waves per SIMD, 16 per CU); workgroup b lands on XCD b % 8, so consecutive envs spread over all eight L2s and
 * every XCD keeps its own copy of the read-only model block.
 */
#include <cstdio>
#include <cstdlib>
#include <gq_device.h>
#include "gq_step_body.h"

/* Parallel build (csrc/Makefile): this file is compiled once per PART, each translation unit instantiating one group of step-kernel
 * variants (-DGQ_PART=k; the variants are 60 large, fully inlined kernels - one unit took 7.5 minutes, the parts build side by side in
 * about one).  GQ_PART undefined: everything in one unit (tools/dev_build.sh development builds).  Part GQ_PART_MISC holds the
 * non-template kernels, the launch entry points and the dispatch over the parts. */
#ifndef GQ_PART
#define GQ_PART (-1)
#endif
#define GQ_PART_MISC 18
#define GQ_IN_MISC (GQ_PART < 0 || GQ_PART == GQ_PART_MISC)
constexpr int gq_step_part(int S, int M, bool C, bool B) { return ((S == 0 ? 0 : (C ? 2 : 1)) * 3 + M) * 2 + (B ? 1 : 0); } /* 0 .. 17 */
constexpr int gq_mailbox_part(bool C, bool B) { return 19 + (C ? 2 : 0) + (B ? 1 : 0); }                                   /* 19 .. 22 */

namespace gq {

/* joint-space PD law of the closed-loop rollouts, every operation rounded on its own (the elementwise torch expression
 * kp * (q_des - q) - kd * qd gives the same bits) */
__device__ __forceinline__ float pd_law(float kp, float kd, float qdes, float q, float qd) {
#pragma clang fp contract(off)
  const float e = qdes - q;
  const float up = kp * e;
  const float ud = kd * qd;
  return up - ud;
}
/* the built-in policy's action for joint j of env `gid` at step `k` of the rollout: the PD law + exploration noise */
__device__ __forceinline__ float pd_action(const GQ_MODEL PolicyPdDev& P, const int j, const float q, const float qd, const int k, const uint32_t gid) {
#pragma clang fp contract(off)
  float u = pd_law(P.kp[j], P.kd[j], P.qdes[j], q, qd);
  if (P.sigma > 0.0f) {
    const float z = philox_normal((uint32_t)j, (uint32_t)(P.step0 + k), gid, 0x9011u, P.seed_lo, P.seed_hi);
    const float nz = P.sigma * z;
    u = u + nz;
  }
  return u;
}
/* inline mode of the closed-loop rollout: lanes 0-11 turn the observation row this wavefront published at the end of its previous
 * step (global memory, the batch's own layout) into the control of the step that starts now; replaces what load_rows left in W.ctrl */
__device__ __forceinline__ void pd_inline(const GQ_MODEL PolicyPdDev& P, const StepArgs& a, const StepCall& c, WaveMem& W, const int env, const int kstep,
                                          const bool apply /* false: the env spends this step on its re-spawn and ignores the action */) {
  const int lane = lane_id();
  if (lane < 12) {
    const int od = mptr(a.batch)->obs_dim;
    const GQ_GLOBAL float* row = gptr(a.obs) + (size_t)env * od;
    const float u = pd_action(P, lane, row[P.col_q[lane]], row[P.col_qd[lane]], kstep, (uint32_t)(env + P.env_id_offset));
    if (apply) W.ctrl[lane] = u;
    if (c.act_seq) gptr(c.act_seq)[((size_t)kstep * a.n_envs + env) * 12 + lane] = u;
  }
}

/* step (+ in-kernel auto-reset).  Same-step mode: a terminated env is re-spawned by the same wavefront - reset_wave,
 * then the reset's own mj_step as a second pass through step_wave; no extra launches, but the launch lasts as long as
 * its two-pass waves.  Next-step mode: the env waits (pending flag) and spends its next launch on reset_wave + the
 * reset's mj_step instead of a user step - every wave runs exactly one mj_step per launch. */
template <int SOLVER, int MODE, bool CONE, bool BOXES, bool SELF, bool PRIM, bool PERSIST = false>
__global__ void __launch_bounds__(GQ_WAVE * GQ_WPB, 4) step_kernel(const FusedArgs* __restrict__ A, const StepCall c) {
  const long long t_entry = (GQ_TICKSET && MODE == 1) ? cycles() : 0; /* sub-stage builds (gq_step_kernel.h GQ_TICKSET) count from here */
  const int widx = wave_index();
  if (GQ_WPB > 1 && widx >= c.count) return;
  const int env = widx + c.env0;
  if (c.mask && !gptr(c.mask)[env]) return; /* wave-uniform */
  __shared__ WaveMem Ws[GQ_WPB];
  WaveMem& W = Ws[GQ_WPB == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))];
#if GQ_TICKSET
  if (lane_id() == 0) W.tk_T = nullptr;
#endif
  for (int kstep = 0;;) { /* one trip, except in a persistent rollout (PERSIST variants, StepCall::n_steps; a variant of their own:
                           * merely compiling the loop in cost the single-step kernel 2.7 %) */
  StepCall ck = c;
  if constexpr (PERSIST) { /* wave-uniform */
    ck.ctrl = c.ctrl ? c.ctrl + (size_t)kstep * c.ctrl_stride : nullptr; /* NULL: inline policy (or zero control) */
    if (c.obs_seq) ck.obs_seq = c.obs_seq + (size_t)kstep * A->s.n_envs * mptr(A->s.batch)->obs_dim;
  }
  const StepCall& c = ck; /* the body below sees this step's call */
  int pass = c.first_pass;
  /* the flags and the env's rows are fetched together: one memory round trip in front of the step (a respawning env - rare -
   * throws the rows away and fetches the ones reset_wave wrote) */
  WaveCtx C;
  int hint = load_rows<SOLVER>(A->s, c, W, env, pass == 0, C);
  bool respawn = c.auto_reset == 2 && C.pend; /* wave-uniform */
  /* (the reset's own step after an explicit gq_reset: the reset kernel left word whether the lift loop is still due - load_rows put it in W.lift_due) */
  if constexpr (PERSIST) if (c.policy) pd_inline(*mptr(c.policy), A->s, c, W, env, kstep, !respawn); /* wave-uniform */
  for (;;) { /* one call site each for reset_wave / step_wave: both are large and fully inlined */
    if (respawn) {
      wave_priority(3); /* reset + step in one launch: this wave is the longest of its SIMD */
      wave_barrier();   /* the rows just staged in LDS are dead: reset_wave reuses the region */
      reset_wave<BOXES, PRIM>(A->r, W, c.env0);
      pass = c.auto_reset;
      hint = load_rows<SOLVER>(A->s, c, W, env, false, C, true);
    }
    const int term = step_wave<SOLVER, MODE, CONE, BOXES, SELF, PRIM>(A->s, c, W, pass, hint, C, t_entry);
    if (pass != 0 || c.auto_reset != 1 || !term) break;
    respawn = true;
  }
  if (!PERSIST || ++kstep >= c.n_steps) break;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); /* the next step reads the rows this one stored (same wave, same addresses) */
  wave_barrier();
  }
}

/* Closed-loop persistent rollout, the stepping side (protocol: gq_step_kernel.h MailboxDev).  grid = any number of one-wave workgroups:
 * each pops tickets of ITS XCD's ready queue until every env-step of the rollout has been claimed.  Production Newton variants only. */
template <int SOLVER, bool CONE, bool BOXES, bool SELF, bool PRIM>
__global__ void __launch_bounds__(GQ_WAVE, 4) mailbox_step_kernel(const FusedArgs* __restrict__ A, const StepCall c0, const MailboxDev* __restrict__ MBp) {
  __shared__ WaveMem W; /* one wavefront per workgroup (the GQ_WPB > 1 experiment builds never launch this kernel) */
#if GQ_TICKSET
  if (lane_id() == 0) W.tk_T = nullptr;
#endif
  const GQ_MODEL MailboxDev& MB = *mptr(MBp);
  const int q = MB.xcc_queue[xcc_id()];
  const int N = MB.n_envs, nq = MB.nq, qmask = MB.qcap - 1, qshift = __builtin_ctz(MB.qcap);
  const int total = ((N - q + nq - 1) / nq) * MB.n_steps; /* env-steps that will ever pass through this queue (< 2^31: checked by gq_rollout_closed) */
  int32_t* const head = MB.q_ctr + (size_t)(3 * q) * GQ_MB_QSTRIDE;
  int32_t* const items = MB.q_items + (size_t)q * MB.qcap;
  int played = 0;
  for (;;) {
    int ticket = 0;
    if (lane_id() == 0) ticket = add_pub(head, 1);
    ticket = __builtin_amdgcn_readfirstlane(ticket);
    if (ticket >= total) break;
    int32_t* const slot = items + (ticket & qmask);
    /* an item carries the LAP of its push ticket (bits 24..30 = (s / qcap) mod 128): tickets may run ahead of the pushes by the number
     * of resident step wavefronts, which can exceed qcap for a small batch - ticket t and ticket t + qcap then wait on the same slot,
     * and each must take the item of ITS lap only (an env stepped by two wavefronts at once otherwise).  Nothing is cleared: the
     * slot is simply overwritten one lap later, and a lap's item is at most one lap old when its ticket reads it. */
    const int want = (ticket >> qshift) & 0x7f;
    int item;
    const long long t0 = wall_clock64();
    for (int spin = 0;; spin++) { /* the ticket's env is pushed as soon as the policy has its action */
      item = __builtin_amdgcn_readfirstlane(ld_pub(slot));
      if ((item & 0xffffff) != 0 && (item >> 24) == want) break;
      nap();
      if ((spin & 31) == 31) {
        bool leave = ld_pub(MB.status) != 0;
        if (!leave && wall_clock64() - t0 > MB.timeout_ticks) { if (lane_id() == 0) { st_pub(MB.status + 1, ticket); st_pub(MB.status, 1); } leave = true; }
        if (leave) { if (lane_id() == 0 && played) add_pub(MB.status + 2, played); return; }
      }
    }
    const int env = (item & 0xffffff) - 1;
    adopt_fence();
    if (MB.flags & 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    /* the env's step index is only needed to place the row in obs_seq: otherwise that round trip is not taken */
    const int k = MB.obs_seq ? __builtin_amdgcn_readfirstlane(ld_pub(MB.steps_done + env)) : 0;
    StepCall ck = c0;
    ck.env0 = env - (int)blockIdx.x; /* step_wave / reset_wave address env0 + wave index */
    ck.ctrl = MB.act;
    ck.obs_seq = MB.obs_seq ? MB.obs_seq + (size_t)k * N * mptr(A->s.batch)->obs_dim : nullptr;
    const StepCall& c = ck;
    int pass = 0;
    if ((MB.flags & 32) && lane_id() == 0) add_pub(MB.issued + N + env, 1 << (4 * xcc_id())); /* experiment: which XCDs ever stepped this env (nibble counters, <= 15 steps) */
    WaveCtx C;
    int hint = load_rows<SOLVER, true>(A->s, c, W, env, true, C);
    bool respawn = c.auto_reset == 2 && C.pend; /* wave-uniform */
    if (respawn) {
      wave_priority(3);
      wave_barrier();
      reset_wave<BOXES, PRIM, true>(A->r, W, c.env0);
      pass = c.auto_reset;
      hint = load_rows<SOLVER, true>(A->s, c, W, env, false, C, true);
    }
    step_wave<SOLVER, 0, CONE, BOXES, SELF, PRIM, true>(A->s, c, W, pass, hint, C);
    publish_fence(); /* state rows are in this XCD's L2, the observation row has been written through */
    if (MB.flags & 4) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (lane_id() == 0) add_pub(MB.steps_done + env, 1);
    wave_barrier();
    played++;
  }
  if (lane_id() == 0 && played) add_pub(MB.status + 2, played);
}

#if GQ_IN_MISC
/* the built-in policy of the closed-loop rollout.  A policy wavefront serves envs of ITS XCD's queue only (lane = env, strided over the
 * policy wavefronts that landed on the XCD): observation row, action row, count and queue slot of an env are written and read
 * through one L2, like the env's state rows - no hand-off of the rollout depends on coherence between two XCDs' L2s, and none
 * needs a device-scope fence (a device-scope acquire on the stepping side costs 14 % of the throughput: it empties the XCD's L2). */
__global__ void __launch_bounds__(GQ_WAVE) policy_pd_kernel(const MailboxDev* __restrict__ MBp, const PolicyPdDev* __restrict__ Pp, const float* __restrict__ obs, const int od) {
  const GQ_MODEL MailboxDev& MB = *mptr(MBp);
  const GQ_MODEL PolicyPdDev& P = *mptr(Pp);
  const int lane = (int)threadIdx.x, nq = MB.nq, q = MB.xcc_queue[xcc_id()], qmask = MB.qcap - 1, qshift = __builtin_ctz(MB.qcap);
  const int N = MB.n_envs, K = MB.n_steps, P_all = (int)gridDim.x;
  int rank = 0;
  if (lane == 0) rank = add_pub(MB.q_ctr + (size_t)(3 * q + 2) * GQ_MB_QSTRIDE, 1);
  rank = __builtin_amdgcn_readfirstlane(rank);
  if (lane == 0) __hip_atomic_fetch_add(MB.alive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  long long t_last = wall_clock64();
  int mine = 0;
  for (;;) { /* every policy wavefront has said where it runs: the stride of this XCD is known */
    int sum = 0;
    for (int x = 0; x < nq; x++) { const int c = ld_pub(MB.q_ctr + (size_t)(3 * x + 2) * GQ_MB_QSTRIDE); sum += c; if (x == q) mine = c; }
    if (sum >= P_all) {
      for (int x = 0; x < nq; x++)
        if (ld_pub(MB.q_ctr + (size_t)(3 * x + 2) * GQ_MB_QSTRIDE) == 0) { if (lane == 0) { st_pub(MB.status + 1, x); st_pub(MB.status, 4); } return; } /* an XCD without policy */
      break;
    }
    nap();
    if (ld_pub(MB.status) != 0) return;
    if (wall_clock64() - t_last > MB.timeout_ticks) { if (lane == 0) { st_pub(MB.status + 1, -1 - rank); st_pub(MB.status, 2); } return; }
  }
  const int nmine = (N - q + nq - 1) / nq;           /* envs of this queue: e = q + nq * i */
  const int g = rank * GQ_WAVE + lane, G = mine * GQ_WAVE;
  int32_t* const tail = MB.q_ctr + (size_t)(3 * q + 1) * GQ_MB_QSTRIDE;
  int32_t* const items = MB.q_items + (size_t)q * MB.qcap;
  t_last = wall_clock64();
  for (;;) {
    bool all_done = true, progress = false;
    for (int i = g; i < nmine; i += G) {
      const int e = q + nq * i;
      const int k = MB.issued[e];
      if (k >= K) continue;
      all_done = false;
      if (ld_pub(MB.steps_done + e) < k) continue; /* the observation after step k - 1 is not out yet */
      if (MB.flags & 8) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      /* (the observation words are read AFTER the count has been seen: loads of one batch may be served in any order) */
      const float* row = obs + (size_t)e * od;
      float qj[12], qd[12];
#pragma unroll
      for (int j = 0; j < 12; j++) { qj[j] = ld_pub(row + P.col_q[j]); qd[j] = ld_pub(row + P.col_qd[j]); }
      /* the action row and the push ticket travel together */
      const int s = add_pub(tail, 1);
#pragma unroll
      for (int j = 0; j < 12; j++) {
        const float a = pd_action(P, j, qj[j], qd[j], k, (uint32_t)(e + P.env_id_offset));
        st_pub(MB.act + (size_t)e * 12 + j, a);
        if (MB.act_seq) MB.act_seq[((size_t)k * N + e) * 12 + j] = a;
      }
      publish_fence();
      if (MB.flags & 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      st_pub(items + (s & qmask), (((s >> qshift) & 0x7f) << 24) | (e + 1)); /* the item (lap of its ticket | env + 1) - after the action is in place */
      MB.issued[e] = k + 1;
      progress = true;
    }
    if (all_done) break;
    if (progress) { t_last = wall_clock64(); continue; }
    nap();
    if (ld_pub(MB.status) != 0) break;
    if (wall_clock64() - t_last > MB.timeout_ticks) { st_pub(MB.status + 1, g); st_pub(MB.status, 2); break; }
  }
}

/* which XCDs does this device expose?  bit HW_REG_XCC_ID of *mask is set by every workgroup */
__global__ void xcc_probe_kernel(int32_t* mask) {
  if (threadIdx.x == 0) atomicOr(mask, 1 << xcc_id());
}

#endif /* GQ_IN_MISC */
template <bool BOXES>
__global__ void __launch_bounds__(GQ_WAVE * GQ_WPB) reset_kernel(ResetArgs a, const int n_envs) {
  const int widx = wave_index();
  if (GQ_WPB > 1 && widx >= n_envs) return;
  if (a.mask && !gptr(a.mask)[widx]) return;
  __shared__ WaveMem Ws[GQ_WPB];
  WaveMem& W = Ws[GQ_WPB == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))];
#if GQ_TICKSET
  if (lane_id() == 0) W.tk_T = nullptr;
#endif
  reset_wave<BOXES>(a, W);
}

#if GQ_IN_MISC
/* HeightMap rays: one wavefront per env, lane = cell (strided when the grid has more than 64).  Scene = the floor plane z = 0
 * plus the static world boxes (mj_ray against static geoms, heightmap.py:90-99): the nearest hit of the vertical ray with any
 * box (slab test in the box frame).  The env's whole grid lies within a circle around `center`: lane = box first picks the
 * boxes whose bounding circle meets it (two ballots), the rays then walk those few instead of every box of the scene (a
 * thread per ray looping over the 100 boxes of random_boxes took 21.7 us per step of config 5). */
__global__ void __launch_bounds__(GQ_WAVE) heightmap_kernel(const GQ_GLOBAL GqDevModel* model, const double* center, int center_stride, const float* yaw, int yaw_stride, int n_envs, int rows, int cols,
                                 float dist_x, float dist_y, float* out) {
  const int env = (int)blockIdx.x;
  center += (size_t)env * center_stride; yaw += (size_t)env * yaw_stride; /* row strides in elements: views of qpos / the observation row work in place */
  heightmap_rays(*model, center[0], center[1], center[2], cosf(yaw[0]), sinf(yaw[0]), rows, cols, dist_x, dist_y, out + (size_t)env * rows * cols * 3);
}

/* mj_jac for one world point per env (include/gq.h gq_jac): kinematics of the env's pose, then lane = dof writes its
 * column: free-joint translations e_k, rotations (base axis a) a x (p - base), hinge on the body's chain axis x (p - anchor). */
__global__ void __launch_bounds__(GQ_WAVE) jac_kernel(const GqDevModel* model, const double* qpos, int body, const double* point, float* jacp, float* jacr) {
  __shared__ WaveMem W;
  const int lane = lane_id(), env = (int)blockIdx.x;
#if GQ_TICKSET
  if (lane == 0) W.tk_T = nullptr;
#endif
  const GQ_MODEL GqDevModel& m = *mptr(model);
  double bxy = 0.0;
  if (lane < 19) {
    const double q = qpos[(size_t)env * 19 + lane];
    if (lane < 2) W.bxy[lane] = q;
    else if (lane == 2) W.basez = (float)q;
    else if (lane < 7) W.qb[lane - 3] = (float)q;
    else W.qj[lane - 7] = (float)q;
  }
  (void)bxy;
  wave_barrier();
  stage_kinematics(W, link_fetch(m, lane));
  /* the point relative to the base x/y (f64 first, like everything else) */
  const V3 p = v3((float)(point[(size_t)env * 3] - W.bxy[0]), (float)(point[(size_t)env * 3 + 1] - W.bxy[1]), (float)point[(size_t)env * 3 + 2]);
  if (lane < GQ_NVD) {
    const int kb = body - 1; /* kernel body index: 0 base, 1 + 3 leg + link */
    V3 jp = v3(0.0f, 0.0f, 0.0f), jr = v3(0.0f, 0.0f, 0.0f);
    if (lane < 3) jp = v3(lane == 0, lane == 1, lane == 2);
    else if (lane < 6) {
      const float* R = W.xmat[0];
      const V3 ax = v3(R[lane - 3], R[3 + lane - 3], R[6 + lane - 3]);
      jr = ax; jp = cross(ax, p - v3(0.0f, 0.0f, W.basez));
    } else {
      const int j = lane - 6, leg = j / 3, depth = j % 3;
      if (kb > 0 && (kb - 1) / 3 == leg && (kb - 1) % 3 >= depth) {
        const V3 ax = ld3(W.u.dyn.axis[j]);
        jr = ax; jp = cross(ax, p - ld3(W.u.dyn.anchor[j]));
      }
    }
    if (jacp) { float* o = jacp + (size_t)env * 54; o[lane] = jp.x; o[18 + lane] = jp.y; o[36 + lane] = jp.z; }
    if (jacr) { float* o = jacr + (size_t)env * 54; o[lane] = jr.x; o[18 + lane] = jr.y; o[36 + lane] = jr.z; }
  }
}

/* mj_ray against the static geoms (include/gq.h gq_ray): one thread per ray; floor plane z = 0, world boxes (slab test in the
 * box frame), height field (the cells under the ray's ground track are walked, two triangles each). */
__device__ inline bool ray_triangle(const double* o, const double* d, const double* a, const double* b, const double* c, double& t) {
  const double e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
  const double pv[3] = {d[1] * e2[2] - d[2] * e2[1], d[2] * e2[0] - d[0] * e2[2], d[0] * e2[1] - d[1] * e2[0]};
  const double det = e1[0] * pv[0] + e1[1] * pv[1] + e1[2] * pv[2];
  if (fabs(det) < 1e-14) return false;
  const double inv = 1.0 / det, tv[3] = {o[0] - a[0], o[1] - a[1], o[2] - a[2]};
  const double u = (tv[0] * pv[0] + tv[1] * pv[1] + tv[2] * pv[2]) * inv;
  if (u < -1e-9 || u > 1.0 + 1e-9) return false;
  const double qv[3] = {tv[1] * e1[2] - tv[2] * e1[1], tv[2] * e1[0] - tv[0] * e1[2], tv[0] * e1[1] - tv[1] * e1[0]};
  const double v = (d[0] * qv[0] + d[1] * qv[1] + d[2] * qv[2]) * inv;
  if (v < -1e-9 || u + v > 1.0 + 1e-9) return false;
  t = (e2[0] * qv[0] + e2[1] * qv[1] + e2[2] * qv[2]) * inv;
  return t >= 0.0;
}
__global__ void ray_kernel(const GQ_GLOBAL GqDevModel* model, const double* origin, const float* dir, int total, float* dist_out, int32_t* geom_out) {
  const int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (idx >= total) return;
  const GQ_GLOBAL GqDevModel& M = *model;
  const double o[3] = {origin[(size_t)idx * 3], origin[(size_t)idx * 3 + 1], origin[(size_t)idx * 3 + 2]};
  const double d[3] = {(double)dir[(size_t)idx * 3], (double)dir[(size_t)idx * 3 + 1], (double)dir[(size_t)idx * 3 + 2]};
  double best = -1.0;
  int geom = -1;
  if (d[2] < 0.0 && o[2] >= 0.0) { best = -o[2] / d[2]; geom = 0; }   /* the floor: a one-sided plane, hit from above */
  for (int b = 0; b < M.nbox; b++) {
    const GQ_GLOBAL GqDevBox& B = M.box[b];
    const double r[3] = {o[0] - (double)B.pos[0], o[1] - (double)B.pos[1], o[2] - (double)B.pos[2]};
    double tin = -1e300, tout = 1e300;
    bool hit = true;
    for (int k = 0; k < 3 && hit; k++) {
      const double ol = (double)B.mat[k] * r[0] + (double)B.mat[3 + k] * r[1] + (double)B.mat[6 + k] * r[2];
      const double dl = (double)B.mat[k] * d[0] + (double)B.mat[3 + k] * d[1] + (double)B.mat[6 + k] * d[2];
      const double s = (double)B.size[k];
      if (fabs(dl) < 1e-14) { hit = fabs(ol) <= s; continue; }
      double t0 = (-s - ol) / dl, t1 = (s - ol) / dl;
      if (t0 > t1) { const double tt = t0; t0 = t1; t1 = tt; }
      if (t0 > tin) tin = t0;
      if (t1 < tout) tout = t1;
      hit = tin <= tout;
    }
    if (!hit || tout < 0.0) continue;
    const double t = tin >= 0.0 ? tin : tout; /* origin inside the box: the exit point, like mju_rayGeom */
    if (best < 0.0 || t < best) { best = t; geom = 1 + b; }
  }
  if (M.hf_nrow > 0) {
    const double sx = M.hf_sx, sy = M.hf_sy, dx = M.hf_dx, dy = M.hf_dy, zmax = (double)M.hf_zmax;
    const double ol[3] = {o[0] - (double)M.hf_pos[0], o[1] - (double)M.hf_pos[1], o[2] - (double)M.hf_pos[2]};
    /* parameter interval of the ray inside the field's bounding box [-sx, sx] x [-sy, sy] x [0, zmax] */
    double t0 = 0.0, t1 = 1e300;
    bool in = true;
    const double lo[3] = {-sx, -sy, 0.0}, hi[3] = {sx, sy, zmax};
    for (int k = 0; k < 3 && in; k++) {
      if (fabs(d[k]) < 1e-14) { in = ol[k] >= lo[k] && ol[k] <= hi[k]; continue; }
      double a = (lo[k] - ol[k]) / d[k], b2 = (hi[k] - ol[k]) / d[k];
      if (a > b2) { const double tt = a; a = b2; b2 = tt; }
      if (a > t0) t0 = a;
      if (b2 < t1) t1 = b2;
      in = t0 <= t1;
    }
    if (in) {
      const int nc = M.hf_ncol, nr = M.hf_nrow;
      const float* H = M.hf_data;
      /* walk the cells along the ground track from t0 to t1 (2-D DDA) */
      double t = t0;
      const double px = ol[0] + t * d[0], py = ol[1] + t * d[1];
      int c = (int)floor((px + sx) / dx), r = (int)floor((py + sy) / dy);
      c = c < 0 ? 0 : (c > nc - 2 ? nc - 2 : c); r = r < 0 ? 0 : (r > nr - 2 ? nr - 2 : r);
      const int stc = d[0] > 0 ? 1 : -1, str = d[1] > 0 ? 1 : -1;
      for (int it = 0; it < nc + nr + 2; it++) {
        const double x0 = -sx + dx * c, y0 = -sy + dy * r, x1 = x0 + dx, y1 = y0 + dy;
        const double h00 = H[r * nc + c], h10 = H[r * nc + c + 1], h01 = H[(r + 1) * nc + c], h11 = H[(r + 1) * nc + c + 1];
        const double A[3] = {x0, y0, h00}, B[3] = {x1, y0, h10}, Cc[3] = {x0, y1, h01}, D[3] = {x1, y1, h11};
        double th, tb = -1.0;
        if (ray_triangle(ol, d, A, B, Cc, th)) tb = th;
        if (ray_triangle(ol, d, D, Cc, B, th) && (tb < 0.0 || th < tb)) tb = th;
        if (tb >= 0.0) { if (best < 0.0 || tb < best) { best = tb; geom = 1 + M.nbox; } break; }
        /* next cell: the nearer of the two cell borders the track crosses */
        const double tx = fabs(d[0]) < 1e-14 ? 1e300 : ((stc > 0 ? x1 : x0) - ol[0]) / d[0];
        const double ty = fabs(d[1]) < 1e-14 ? 1e300 : ((str > 0 ? y1 : y0) - ol[1]) / d[1];
        if (tx < ty) { c += stc; t = tx; } else { r += str; t = ty; }
        if (t > t1 || c < 0 || r < 0 || c > nc - 2 || r > nr - 2) break;
      }
    }
  }
  dist_out[idx] = (float)best;
  if (geom_out) geom_out[idx] = geom;
}

#endif /* GQ_IN_MISC */
}  // namespace gq

#if GQ_IN_MISC
extern "C" void gq_launch_heightmap(const GQ_GLOBAL GqDevModel* model, const double* center, int center_stride, const float* yaw, int yaw_stride, int n_envs, int rows, int cols,
                                    float dist_x, float dist_y, float* out, hipStream_t stream) {
  hipLaunchKernelGGL(gq::heightmap_kernel, dim3(n_envs), dim3(GQ_WAVE), 0, stream, model, center, center_stride, yaw, yaw_stride, n_envs, rows, cols, dist_x, dist_y, out);
}

#endif /* GQ_IN_MISC */
/* Development builds (tools/dev_build.sh: -DGQ_DEV_ONLY=<0|1>, cone = 0 pyramidal / 1 elliptic) instantiate only the flat-scene
 * self-collision Newton variants (production + instrumented) - a 15 s build for A/B timing of kernel experiments through
 * GQ_LIBGQ_PATH; any other launch aborts.  The product library is built without the macro and carries every variant. */
#ifndef GQ_DEV_BOXES
#define GQ_DEV_BOXES 0 /* -DGQ_DEV_BOXES=1: the development build carries the world-box / height-field variants instead of the flat ones */
#endif
#ifndef GQ_DEV_CUTS
#define GQ_DEV_CUTS 0 /* -DGQ_DEV_CUTS=1: the development build also carries the stage-cut variant (tools/stage_cuts.py) */
#endif
template <int S, int M, bool C, bool B, bool SF, bool P = true>
static bool launch_variant(const gq::FusedArgs* dev_args, const gq::StepCall* c, int n_envs, hipStream_t stream) {
  if constexpr (GQ_PART >= 0 && gq_step_part(S, M, C, B) != GQ_PART) return false; else
#ifdef GQ_DEV_ONLY
  if constexpr (!(S == 1 && (M != 2 || GQ_DEV_CUTS) && B == (GQ_DEV_BOXES != 0) && (!B || P == (GQ_DEV_BOXES == 2)) && SF && C == (GQ_DEV_ONLY != 0))) { fprintf(stderr, "libgq development build: kernel variant solver=%d mode=%d cone=%d boxes=%d self=%d not compiled in\n", S, M, int(C), int(B), int(SF)); abort(); } else
#endif
  {
    gq::StepCall call = *c;
    call.count = n_envs;
    if constexpr (M == 0) {
      if (c->n_steps > 1 || c->policy) { /* persistent rollout (also a one-step one with the policy inline: only this variant evaluates it): production kernel only */
        hipLaunchKernelGGL((gq::step_kernel<S, M, C, B, SF, P, true>), dim3((n_envs + GQ_WPB - 1) / GQ_WPB), dim3(GQ_WAVE * GQ_WPB), 0, stream, dev_args, call);
        return true;
      }
    }
    call.n_steps = 1;
    hipLaunchKernelGGL((gq::step_kernel<S, M, C, B, SF, P>), dim3((n_envs + GQ_WPB - 1) / GQ_WPB), dim3(GQ_WAVE * GQ_WPB), 0, stream, dev_args, call);
    return true;
  }
}

/* the run-time choice among the variants of THIS translation unit; false: the variant lives in another part */
static bool dispatch_step(const gq::FusedArgs* dev_args, const gq::StepCall* c, int n_envs, int solver, int cone, int boxes, int self, hipStream_t stream) {
  /* 0: production; 1: debug record + stage timers; 2: stage cut (GQ_STOP_STAGE / gq_debug_stop_stage) - the early returns
   * of the cut cost the production kernel ~8 % when merely compiled in, hence a variant of their own.
   * Scene variants: flat (no world geoms beyond the floor), flat + robot self-collision, world boxes / height field (always
   * with the self-collision stage compiled in; a model without pairs skips it at run time). */
  const int mode = c->debug != nullptr ? 1 : (c->stop_stage != 0 ? 2 : 0);
#define GQ_LAUNCH(S, M, C) (boxes == 2 ? launch_variant<S, M, C, true, true, true>(dev_args, c, n_envs, stream) \
                            : boxes ? launch_variant<S, M, C, true, true, false>(dev_args, c, n_envs, stream) \
                            : self ? launch_variant<S, M, C, false, true>(dev_args, c, n_envs, stream) \
                            : launch_variant<S, M, C, false, false>(dev_args, c, n_envs, stream))
#define GQ_LAUNCH_MODE(S, C) (mode == 1 ? GQ_LAUNCH(S, 1, C) : mode == 2 ? GQ_LAUNCH(S, 2, C) : GQ_LAUNCH(S, 0, C))
  if (solver == 1 && cone) return GQ_LAUNCH_MODE(1, true);
  if (solver == 1) return GQ_LAUNCH_MODE(1, false);
  return GQ_LAUNCH_MODE(0, false); /* PGS: pyramidal cones only (gq_model_create rejects elliptic cones with solver 0) */
#undef GQ_LAUNCH_MODE
#undef GQ_LAUNCH
}
#define GQ_CAT2(a, b) a##b
#define GQ_CAT(a, b) GQ_CAT2(a, b)
#if GQ_PART >= 0 && GQ_PART < GQ_PART_MISC
extern "C" bool GQ_CAT(gq_launch_step_p, GQ_PART)(const gq::FusedArgs* dev_args, const gq::StepCall* c, int n_envs, int solver, int cone, int boxes, int self, hipStream_t stream) {
  return dispatch_step(dev_args, c, n_envs, solver, cone, boxes, self, stream);
}
#endif
#if GQ_IN_MISC
#if GQ_PART >= 0
#define GQ_P(k) extern "C" bool gq_launch_step_p##k(const gq::FusedArgs*, const gq::StepCall*, int, int, int, int, int, hipStream_t);
GQ_P(0) GQ_P(1) GQ_P(2) GQ_P(3) GQ_P(4) GQ_P(5) GQ_P(6) GQ_P(7) GQ_P(8) GQ_P(9) GQ_P(10) GQ_P(11) GQ_P(12) GQ_P(13) GQ_P(14) GQ_P(15) GQ_P(16) GQ_P(17)
#undef GQ_P
#endif
extern "C" void gq_launch_step(const gq::FusedArgs* dev_args, const gq::StepCall* c, int n_envs, int solver, int cone, int boxes, int self, hipStream_t stream) {
#if GQ_PART >= 0
  static bool (*const part[18])(const gq::FusedArgs*, const gq::StepCall*, int, int, int, int, int, hipStream_t) = {
    gq_launch_step_p0, gq_launch_step_p1, gq_launch_step_p2, gq_launch_step_p3, gq_launch_step_p4, gq_launch_step_p5, gq_launch_step_p6, gq_launch_step_p7, gq_launch_step_p8,
    gq_launch_step_p9, gq_launch_step_p10, gq_launch_step_p11, gq_launch_step_p12, gq_launch_step_p13, gq_launch_step_p14, gq_launch_step_p15, gq_launch_step_p16, gq_launch_step_p17};
  const int mode = c->debug != nullptr ? 1 : (c->stop_stage != 0 ? 2 : 0);
  const int k = gq_step_part(solver == 1 ? 1 : 0, mode, solver == 1 && cone, boxes != 0);
  if (!part[k](dev_args, c, n_envs, solver, cone, boxes, self, stream)) { fprintf(stderr, "libgq: step-kernel part %d does not hold solver=%d mode=%d cone=%d boxes=%d self=%d\n", k, solver, mode, cone, boxes, self); abort(); }
#else
  dispatch_step(dev_args, c, n_envs, solver, cone, boxes, self, stream);
#endif
}
extern "C" void gq_launch_xcc_probe(int32_t* mask, hipStream_t stream) {
  hipLaunchKernelGGL(gq::xcc_probe_kernel, dim3(4096), dim3(GQ_WAVE), 0, stream, mask);
}
extern "C" void gq_launch_policy_pd(const gq::MailboxDev* mb, const gq::PolicyPdDev* pd, const float* obs, int od, int waves, hipStream_t stream) {
  hipLaunchKernelGGL(gq::policy_pd_kernel, dim3(waves), dim3(GQ_WAVE), 0, stream, mb, pd, obs, od);
}
#endif /* GQ_IN_MISC */
/* mailbox variants of this translation unit (parts 19 .. 22: pyramidal / elliptic x flat / world geoms) */
template <bool C, bool B, bool SF, bool P>
static bool launch_mailbox_variant(const gq::FusedArgs* dev_args, const gq::StepCall* c, const gq::MailboxDev* mb, int waves, hipStream_t stream) {
  if constexpr (GQ_PART >= 0 && gq_mailbox_part(C, B) != GQ_PART) return false;
  else { hipLaunchKernelGGL((gq::mailbox_step_kernel<1, C, B, SF, P>), dim3(waves), dim3(GQ_WAVE), 0, stream, dev_args, *c, mb); return true; }
}
static bool dispatch_mailbox(const gq::FusedArgs* dev_args, const gq::StepCall* c, const gq::MailboxDev* mb, int waves, int cone, int boxes, int self, hipStream_t stream) {
#if GQ_WPB != 1
  return false;
#elif defined(GQ_DEV_ONLY)
  if (!(!boxes && self && cone == (GQ_DEV_ONLY != 0))) return false;
  return launch_mailbox_variant<(GQ_DEV_ONLY != 0), false, true, true>(dev_args, c, mb, waves, stream);
#else
#define GQ_MB_SCENE(C) (boxes == 2 ? launch_mailbox_variant<C, true, true, true>(dev_args, c, mb, waves, stream) \
                        : boxes ? launch_mailbox_variant<C, true, true, false>(dev_args, c, mb, waves, stream) \
                        : self ? launch_mailbox_variant<C, false, true, true>(dev_args, c, mb, waves, stream) \
                        : launch_mailbox_variant<C, false, false, true>(dev_args, c, mb, waves, stream))
  return cone ? GQ_MB_SCENE(true) : GQ_MB_SCENE(false);
#undef GQ_MB_SCENE
#endif
}
#if GQ_PART > GQ_PART_MISC
extern "C" bool GQ_CAT(gq_launch_mailbox_p, GQ_PART)(const gq::FusedArgs* dev_args, const gq::StepCall* c, const gq::MailboxDev* mb, int waves, int cone, int boxes, int self, hipStream_t stream) {
  return dispatch_mailbox(dev_args, c, mb, waves, cone, boxes, self, stream);
}
#endif
#if GQ_IN_MISC
#if GQ_PART >= 0
#define GQ_P(k) extern "C" bool gq_launch_mailbox_p##k(const gq::FusedArgs*, const gq::StepCall*, const gq::MailboxDev*, int, int, int, int, hipStream_t);
GQ_P(19) GQ_P(20) GQ_P(21) GQ_P(22)
#undef GQ_P
#endif
/* returns 0 if the scene / solver combination has no mailbox variant compiled in */
extern "C" int gq_launch_mailbox_step(const gq::FusedArgs* dev_args, const gq::StepCall* c, const gq::MailboxDev* mb, int waves, int solver, int cone, int boxes, int self, hipStream_t stream) {
  if (solver != 1) return 0;
#if GQ_PART >= 0
  static bool (*const part[4])(const gq::FusedArgs*, const gq::StepCall*, const gq::MailboxDev*, int, int, int, int, hipStream_t) = {
    gq_launch_mailbox_p19, gq_launch_mailbox_p20, gq_launch_mailbox_p21, gq_launch_mailbox_p22};
  return part[gq_mailbox_part(cone != 0, boxes != 0) - 19](dev_args, c, mb, waves, cone, boxes, self, stream) ? 1 : 0;
#else
  return dispatch_mailbox(dev_args, c, mb, waves, cone, boxes, self, stream) ? 1 : 0;
#endif
}
extern "C" void gq_launch_jac(const GqDevModel* model, const double* qpos, int body, const double* point, float* jacp, float* jacr, int n_envs, hipStream_t stream) {
  hipLaunchKernelGGL(gq::jac_kernel, dim3(n_envs), dim3(GQ_WAVE), 0, stream, model, qpos, body, point, jacp, jacr);
}
extern "C" void gq_launch_ray(const GQ_GLOBAL GqDevModel* model, const double* origin, const float* dir, int total, float* dist, int32_t* geom, hipStream_t stream) {
  hipLaunchKernelGGL(gq::ray_kernel, dim3((total + 127) / 128), dim3(128), 0, stream, model, origin, dir, total, dist, geom);
}
extern "C" void gq_launch_reset(const gq::ResetArgs* a, int n_envs, int boxes, hipStream_t stream) {
  if (boxes) hipLaunchKernelGGL(gq::reset_kernel<true>, dim3((n_envs + GQ_WPB - 1) / GQ_WPB), dim3(GQ_WAVE * GQ_WPB), 0, stream, *a, n_envs);
  else hipLaunchKernelGGL(gq::reset_kernel<false>, dim3((n_envs + GQ_WPB - 1) / GQ_WPB), dim3(GQ_WAVE * GQ_WPB), 0, stream, *a, n_envs);
}
#endif /* GQ_IN_MISC */
