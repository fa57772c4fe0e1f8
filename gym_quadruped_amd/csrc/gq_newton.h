/*
 * gq_newton.h - primal Newton solver of the constraint problem (MuJoCo's default solver, mj_solNewton), one env per
 * wavefront.  Restated on the CPU in oracle/gq_oracle.c::gqo_sol_newton.
 *
 *   minimise over qacc:  1/2 (qacc - qacc_smooth)' M (qacc - qacc_smooth) + sum_i s_i(J_i qacc - aref_i)
 *
 * lane = constraint row (J row, R, aref, bounds in registers), lane = dof for the 18-vectors, lanes 0-3 = legs for
 * the tree-sparse factorisation.  The Hessian H = M + J' diag(D active) J has exactly M's tree sparsity (every row
 * of J touches the base and at most one leg), so it reuses the L'DL machinery; no nefc x nefc dual operator is
 * ever formed.
 */
#pragma once
#include "gq_step_kernel.h"

namespace gq {

/* out[d] = sum_k M[d][k] v[k], lane d < 18, v and out in LDS (tree-sparse M) */
__device__ __forceinline__ float mul_m_row(const WaveMem& W, const float* v, int d) {
  float s = 0.0f;
  if (d < 6) {
#pragma unroll
    for (int k = 0; k < 6; k++) s += W.Mb[d][k] * v[k];
#pragma unroll
    for (int j = 0; j < GQ_NJ; j++) s += W.Mc[j][d] * v[6 + j];
  } else {
    const int j = d - 6, l0 = 3 * (j / 3), dep = j % 3;
#pragma unroll
    for (int k = 0; k < 6; k++) s += W.Mc[j][k] * v[k];
#pragma unroll
    for (int q = 0; q < 3; q++) { /* same-leg block, symmetric: entry (max, min) */
      const int hi = q > dep ? q : dep, lo = q > dep ? dep : q;
      s += W.Mc[l0 + hi][6 + lo] * v[6 + l0 + q];
    }
  }
  return s;
}

/* L'DL of one tree-sparse system given in the Mc/Mb layout -> F (lanes 0-3 legs, lane 0 base block) */
__device__ inline void factor_tree_one(WaveMem& W, const float (*Sc)[9], const float (*Sb)[6], float* F) {
  const int lane = lane_id();
  float(*acc)[21] = W.acc2;
  if (lane < 4) {
    const int hh = 6 + 3 * lane, t = hh + 1, c = hh + 2;
    float rc[9], rt[8], rh[7], bb[21];
#pragma unroll
    for (int j = 0; j < 6; j++) { rc[j] = Sc[c - 6][j]; rt[j] = Sc[t - 6][j]; rh[j] = Sc[hh - 6][j]; }
    rc[6] = Sc[c - 6][6]; rc[7] = Sc[c - 6][7]; rc[8] = Sc[c - 6][8];
    rt[6] = Sc[t - 6][6]; rt[7] = Sc[t - 6][7];
    rh[6] = Sc[hh - 6][6];
#pragma unroll
    for (int q = 0; q < 21; q++) bb[q] = 0.0f;
    const float ic = fast_rcp(rc[8]);
    {
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
    {
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
#pragma unroll
    for (int i = 5; i >= 0; i--) {
      const float tmp = rh[i] * ih;
#pragma unroll
      for (int j = 0; j <= i; j++) bb[i * (i + 1) / 2 + j] -= rh[j] * tmp;
      rh[i] = tmp;
    }
#pragma unroll
    for (int j = 0; j < 6; j++) { F[GQ_F_LC(c - 6, j)] = rc[j]; F[GQ_F_LC(t - 6, j)] = rt[j]; F[GQ_F_LC(hh - 6, j)] = rh[j]; }
    F[GQ_F_LC(c - 6, 6)] = rc[6]; F[GQ_F_LC(c - 6, 7)] = rc[7]; F[GQ_F_LC(t - 6, 6)] = rt[6];
    F[GQ_F_DINV(c)] = ic; F[GQ_F_DINV(t)] = it; F[GQ_F_DINV(hh)] = ih;
#pragma unroll
    for (int q = 0; q < 21; q++) acc[lane][q] = bb[q];
  }
  wave_barrier();
  { /* 6x6 base block, row-parallel: lane i < 6 owns row i (lower part); pivots k = 5..0, row k is broadcast with
     * v_readlane and the rows above it are updated at once */
    float row[6];
#pragma unroll
    for (int j = 0; j < 6; j++) {
      float v = 0.0f;
      if (lane < 6 && j <= lane) {
        const int q = lane * (lane + 1) / 2 + j;
        v = Sb[lane][j] + acc[0][q] + acc[1][q] + acc[2][q] + acc[3][q];
      }
      row[j] = v;
    }
#pragma unroll
    for (int k = 5; k >= 0; k--) {
      float rk[6];
#pragma unroll
      for (int j = 0; j <= k; j++) rk[j] = bcast(row[j], k);
      const float inv = fast_rcp(rk[k]);
      float rki = 0.0f;
#pragma unroll
      for (int j = 0; j < k; j++) rki = (lane == j) ? rk[j] : rki;
      const float tmp = rki * inv;
#pragma unroll
      for (int j = 0; j < k; j++) row[j] -= (lane < k && j <= lane) ? rk[j] * tmp : 0.0f;
      if (lane == k) {
        F[GQ_F_DINV(k)] = inv;
#pragma unroll
        for (int j = 0; j < k; j++) F[GQ_F_LB(k, j)] = rk[j] * inv;
      }
    }
  }
  wave_barrier();
}

/* out <- S^-1 g for a tree-sparse symmetric positive definite S (Mc/Mb layout) + h_d * damping on the diagonal: Gaussian
 * elimination and back-substitution fused, entirely in the registers of lanes 0-3.  Lane L eliminates calf, thigh and hip
 * of leg L from the augmented system (S | g); the four Schur contributions to the 6x6 base block and its right-hand side
 * are summed across the quad with DPP (no LDS, no barrier); every one of the four lanes then solves the same 6x6 system
 * redundantly, so the base solution is already where the back-substitution of each leg needs it.  The factor is never
 * stored: each system is solved for exactly one right-hand side (Newton search direction, qacc_smooth, the Euler
 * system), which is what made storing L and D in LDS, re-reading them and three barriers per solve pure latency.
 * g and out: LDS [18], may alias. */
/* STORE: the second quad of lanes (4-7) eliminates the SAME matrix plus damping[] on its diagonal (the Euler system
 * M + h diag(damping), damping = h * dof_damping in LDS) in the same instruction stream - those lanes are idle anyway - and
 * leaves its factor in LDS: per leg 24 floats (multipliers of calf / thigh / hip rows, the three pivot reciprocals) in fleg[4][24],
 * the base block's 15 multipliers + 6 reciprocals in fbase[21].  solve_tree_stored then solves the Euler system by
 * substitution alone (~100 VALU instead of the ~300 of an elimination). */
#define GQ_EULER_FLEG(W) (&(W).F[1][0])   /* [4][24]; the GEN variants' per-geom hit normals (S6) are dead by the solver */
#define GQ_EULER_FBASE(W) (&(W).F[0][56]) /* [21] */
template <bool DAMP, bool STORE = false>
__device__ __forceinline__ void solve_tree_fused(const float (*Sc)[9], const float (*Sb)[6], const float* damping, float hd,
                                        const float* g, float* out, float* fleg = nullptr, float* fbase = nullptr) {
  const int lane = lane_id();
  { /* every lane runs the same instruction stream (lanes >= 4 mirror leg 0 and are discarded at the end): no branch, and the
     * cross-lane sums are called from wave-uniform control flow */
    const int hh = 3 * (STORE ? (lane < 8 ? (lane & 3) : 0) : (lane < 4 ? lane : 0)), t = hh + 1, c = hh + 2; /* joint indices of the leg (dof = 6 + joint) */
    const bool second = STORE && lane >= 4 && lane < 8;   /* this lane eliminates M + diag(damping) and stores the factor */
    const bool first = STORE ? (lane & 3) == 0 : lane == 0; /* carries the base block of its quad */
    if constexpr (STORE) hd = second ? 1.0f : 0.0f;
    float rc[9], rt[8], rh[7], bb[21], gb[6];
#pragma unroll
    for (int j = 0; j < 6; j++) { rc[j] = Sc[c][j]; rt[j] = Sc[t][j]; rh[j] = Sc[hh][j]; }
    rc[6] = Sc[c][6]; rc[7] = Sc[c][7]; rc[8] = Sc[c][8];
    rt[6] = Sc[t][6]; rt[7] = Sc[t][7];
    rh[6] = Sc[hh][6];
    if constexpr (DAMP || STORE) { rc[8] += hd * damping[6 + c]; rt[7] += hd * damping[6 + t]; rh[6] += hd * damping[6 + hh]; }
    float gc = g[6 + c], gt = g[6 + t], gh = g[6 + hh];
    /* base block and right-hand side: lane 0 carries S_bb and g_b, the others start from zero (summed below) */
#pragma unroll
    for (int i = 0; i < 6; i++) {
      gb[i] = (STORE ? first : lane == 0) ? g[i] : 0.0f;
#pragma unroll
      for (int j = 0; j <= i; j++) bb[i * (i + 1) / 2 + j] = (STORE ? first : lane == 0) ? Sb[i][j] + (((DAMP || STORE) && i == j) ? hd * damping[i] : 0.0f) : 0.0f;
    }
    /* eliminate the calf: rows thigh, hip, base */
    const float ic = fast_rcp(rc[8]);
    {
      float f = rc[7] * ic;
#pragma unroll
      for (int j = 0; j <= 7; j++) rt[j] -= rc[j] * f;
      gt -= gc * f;
      f = rc[6] * ic;
#pragma unroll
      for (int j = 0; j <= 6; j++) rh[j] -= rc[j] * f;
      gh -= gc * f;
#pragma unroll
      for (int i = 0; i < 6; i++) {
        f = rc[i] * ic;
#pragma unroll
        for (int j = 0; j <= i; j++) bb[i * (i + 1) / 2 + j] -= rc[j] * f;
        gb[i] -= gc * f;
      }
    }
    const float it = fast_rcp(rt[7]);
    {
      float f = rt[6] * it;
#pragma unroll
      for (int j = 0; j <= 6; j++) rh[j] -= rt[j] * f;
      gh -= gt * f;
#pragma unroll
      for (int i = 0; i < 6; i++) {
        f = rt[i] * it;
#pragma unroll
        for (int j = 0; j <= i; j++) bb[i * (i + 1) / 2 + j] -= rt[j] * f;
        gb[i] -= gt * f;
      }
    }
    const float ih = fast_rcp(rh[6]);
#pragma unroll
    for (int i = 0; i < 6; i++) {
      const float f = rh[i] * ih;
#pragma unroll
      for (int j = 0; j <= i; j++) bb[i * (i + 1) / 2 + j] -= rh[j] * f;
      gb[i] -= gh * f;
    }
    if constexpr (STORE) if (second) { /* the leg's multipliers and pivot reciprocals */
      float* F = fleg + 24 * (lane & 3);
#pragma unroll
      for (int j = 0; j < 8; j++) F[j] = rc[j] * ic;
#pragma unroll
      for (int j = 0; j < 7; j++) F[8 + j] = rt[j] * it;
#pragma unroll
      for (int j = 0; j < 6; j++) F[15 + j] = rh[j] * ih;
      F[21] = ic; F[22] = it; F[23] = ih;
    }
    /* Schur complement of the base: sum of the four legs */
#pragma unroll
    for (int q = 0; q < 21; q++) bb[q] = quad_sum(bb[q]);
#pragma unroll
    for (int i = 0; i < 6; i++) gb[i] = quad_sum(gb[i]);
    /* 6x6 base system, redundantly on each of the four lanes: elimination from the last row up (same order as the tree) */
    float xb[6];
#pragma unroll
    for (int k = 5; k >= 0; k--) {
      const float inv = fast_rcp(bb[k * (k + 1) / 2 + k]);
#pragma unroll
      for (int i = 0; i < k; i++) {
        const float f = bb[k * (k + 1) / 2 + i] * inv;
#pragma unroll
        for (int j = 0; j <= i; j++) bb[i * (i + 1) / 2 + j] -= bb[k * (k + 1) / 2 + j] * f;
        gb[i] -= gb[k] * f;
      }
      gb[k] *= inv; /* gb[k] now holds (g_k - sum_{j<k} S_kj x_j ... ) / S_kk once the x_j below are known */
      bb[k * (k + 1) / 2 + k] = inv;
      if constexpr (STORE) { /* the multipliers l_ki take the place of the eliminated row (it is not read again) */
#pragma unroll
        for (int i = 0; i < k; i++) bb[k * (k + 1) / 2 + i] *= inv;
      }
    }
    if constexpr (STORE) if (lane == 4) {
#pragma unroll
      for (int q = 0; q < 21; q++) fbase[q] = bb[q]; /* strictly lower entries: multipliers; diagonal: reciprocals */
    }
#pragma unroll
    for (int k = 0; k < 6; k++) {
      float s = gb[k];
#pragma unroll
      for (int j = 0; j < k; j++) s -= (STORE ? bb[k * (k + 1) / 2 + j] : bb[k * (k + 1) / 2 + j] * bb[k * (k + 1) / 2 + k]) * xb[j];
      xb[k] = s;
    }
    /* back-substitution up the leg: hip, thigh, calf */
    float xh = gh, xt = gt, xc = gc;
#pragma unroll
    for (int j = 0; j < 6; j++) xh -= rh[j] * xb[j];
    xh *= ih;
#pragma unroll
    for (int j = 0; j < 6; j++) xt -= rt[j] * xb[j];
    xt = (xt - rt[6] * xh) * it;
#pragma unroll
    for (int j = 0; j < 6; j++) xc -= rc[j] * xb[j];
    xc = (xc - rc[6] * xh - rc[7] * xt) * ic;
    wave_barrier(); /* g may alias out: every lane has read its right-hand side */
    if (lane < 4) { out[6 + hh] = xh; out[6 + t] = xt; out[6 + c] = xc; }
    if (lane == 0) {
#pragma unroll
      for (int j = 0; j < 6; j++) out[j] = xb[j];
    }
  }
  wave_barrier();
}

/* out <- (M + diag(damping))^-1 g by substitution with the factor solve_tree_fused<.., STORE> left in LDS: lanes 0-3 run the
 * forward pass of their leg (the right-hand side through the eliminations of calf, thigh, hip), the four contributions to the
 * base right-hand side are summed across the quad, every lane solves the 6x6 base system from the stored multipliers and
 * substitutes back up its leg.  g and out: LDS [18], may alias. */
__device__ __forceinline__ void solve_tree_stored(const float* fleg, const float* fbase, const float* g, float* out) {
  const int lane = lane_id();
  const int L = lane < 4 ? lane : 0, hh = 3 * L, t = hh + 1, c = hh + 2;
  const float* F = fleg + 24 * L;
  float fc[8], ft[7], fh[6], lb[21], gb[6];
#pragma unroll
  for (int j = 0; j < 8; j++) fc[j] = F[j];
#pragma unroll
  for (int j = 0; j < 7; j++) ft[j] = F[8 + j];
#pragma unroll
  for (int j = 0; j < 6; j++) fh[j] = F[15 + j];
  const float ic = F[21], it = F[22], ih = F[23];
#pragma unroll
  for (int q = 0; q < 21; q++) lb[q] = fbase[q];
  float gc = g[6 + c], gt = g[6 + t], gh = g[6 + hh];
#pragma unroll
  for (int i = 0; i < 6; i++) gb[i] = lane == 0 ? g[i] : 0.0f;
  gt -= gc * fc[7]; gh -= gc * fc[6];
#pragma unroll
  for (int i = 0; i < 6; i++) gb[i] -= gc * fc[i];
  gh -= gt * ft[6];
#pragma unroll
  for (int i = 0; i < 6; i++) gb[i] -= gt * ft[i];
#pragma unroll
  for (int i = 0; i < 6; i++) gb[i] -= gh * fh[i];
#pragma unroll
  for (int i = 0; i < 6; i++) gb[i] = quad_sum(gb[i]);
#pragma unroll
  for (int k = 5; k >= 0; k--) {
#pragma unroll
    for (int i = 0; i < k; i++) gb[i] -= gb[k] * lb[k * (k + 1) / 2 + i];
    gb[k] *= lb[k * (k + 1) / 2 + k];
  }
  float xb[6];
#pragma unroll
  for (int k = 0; k < 6; k++) {
    float sx = gb[k];
#pragma unroll
    for (int j = 0; j < k; j++) sx -= lb[k * (k + 1) / 2 + j] * xb[j];
    xb[k] = sx;
  }
  float xh = gh * ih, xt = gt * it, xc = gc * ic;
#pragma unroll
  for (int j = 0; j < 6; j++) { xh -= fh[j] * xb[j]; xt -= ft[j] * xb[j]; xc -= fc[j] * xb[j]; }
  xt -= ft[6] * xh;
  xc -= fc[6] * xh + fc[7] * xt;
  wave_barrier(); /* g may alias out: every lane has read its right-hand side */
  if (lane < 4) { out[6 + hh] = xh; out[6 + t] = xt; out[6 + c] = xc; }
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < 6; j++) out[j] = xb[j];
  }
  wave_barrier();
}

/* Newton step of an env whose contacts couple two legs (robot self-collision): M + J'DJ has lost the tree sparsity the
 * fused elimination relies on, so the system is solved DENSE, lane-parallel.  Lane i < 18 holds row i of H in registers:
 * the tree-sparse part comes from the Hessian the regular assembly just wrote (Hc / Hb: it walks every row, so base and
 * same-leg entries already contain the coupling rows' share), the entries between two different legs - zero in M - are
 * summed here over the robot-robot rows [r0, r1) only (the only rows that touch two legs; W.force holds the row weights).
 * Then Gaussian elimination without pivoting (H is SPD): the pivot row is broadcast with v_readlane and every lane below
 * the pivot updates its own row at once; back substitution the same way.  Out of line on purpose: its registers are saved
 * at the call site, inside the rare branch, instead of being reserved across the whole Newton loop. */
__device__ __forceinline__ void newton_dense_step(WaveMem& W, int r0, int r1, const float* g, float* out) {
  const int lane = lane_id();
  const int i = lane < GQ_NVD ? lane : 0; /* lanes >= 18 mirror row 0 and are discarded */
  float row[GQ_NVD], b = g[i];
  const int li = i < 6 ? -1 : (i - 6) / 3;
  /* every load below is unconditional (a clamped address, the value masked afterwards): conditional loads compile to one
   * exec-masked branch and one exposed LDS latency EACH - 12 in a row per coupling row made this step cost 28k cycles */
  const float* Hc0 = &W.u2.n.Hc[0][0];
  const float* Hb0 = &W.u2.n.Hb[0][0];
#pragma unroll
  for (int j = 0; j < GQ_NVD; j++) {
    const int hi = i > j ? i : j, lo = i > j ? j : i;
    if (j < 6) { /* compile-time: column in the base block */
      row[j] = hi < 6 ? Hb0[hi * 6 + lo] : Hc0[(hi - 6) * 9 + lo];
    } else {
      const bool same = (hi - 6) / 3 == (lo - 6) / 3;
      const int a_c = (hi - 6) * 9 + (lo < 6 ? lo : 6 + (lo - 6) % 3); /* lo < 6 only when i < 6: base row, leg column */
      const float v = Hc0[hi >= 6 ? a_c : 0];
      row[j] = (lo < 6 || same) ? v : 0.0f;
    }
  }
  /* which rows of [r0, r1) really couple two legs with a non-zero weight?  lane = row looks at its own J row (the range also
   * holds same-leg / leg-trunk contacts, inactive rows and - elliptic cones - the virtual rows of every middle-zone contact:
   * 20-30 rows, of which 3-8 matter) */
  uint64_t xm = (r1 >= 64 ? ~0ull : ((1ull << r1) - 1ull)) & ~((1ull << r0) - 1ull); /* a short range is walked as it is */
  if (r1 - r0 > 4) {
    bool cross = false;
    if (lane >= r0 && lane < r1 && W.force[lane] != 0.0f) {
      const float* J = W.u.B[opaque_lane(lane)];
      int nl = 0;
#pragma unroll
      for (int l = 0; l < 4; l++) nl += (J[6 + 3 * l] != 0.0f || J[7 + 3 * l] != 0.0f || J[8 + 3 * l] != 0.0f) ? 1 : 0;
      cross = nl >= 2;
    }
    xm = ballot(cross);
  }
  while (xm) { /* wave-uniform: cross-leg entries */
    const int r = ffs64(xm);
    xm &= xm - 1;
    float bj[12];
#pragma unroll
    for (int j = 0; j < 12; j++) bj[j] = W.u.B[r][6 + j];
    const float a = li >= 0 ? W.force[r] * W.u.B[r][i] : 0.0f;
#pragma unroll
    for (int j = 0; j < 12; j++) row[6 + j] += (j / 3 != li) ? a * bj[j] : 0.0f;
  }
#pragma unroll
  for (int k = 0; k < GQ_NVD - 1; k++) {
    const float inv = fast_rcp(bcast(row[k], k));
    const float f = lane > k ? row[k] * inv : 0.0f;
    /* the pivot row is broadcast first (v_readlane -> SGPR), the updates follow: interleaved, every FMA waited one s_nop for
     * the SGPR its readlane had just written */
    float pv[GQ_NVD];
    const float pb = bcast(b, k);
#pragma unroll
    for (int j = k + 1; j < GQ_NVD; j++) pv[j] = bcast(row[j], k);
    b -= f * pb;
#pragma unroll
    for (int j = k + 1; j < GQ_NVD; j++) row[j] -= f * pv[j];
  }
  float x = 0.0f;
#pragma unroll
  for (int k = GQ_NVD - 1; k >= 0; k--) {
    const float xk = bcast(b, k) * fast_rcp(bcast(row[k], k));
    x = lane == k ? xk : x;
    b -= lane < k ? row[k] * xk : 0.0f;
  }
  wave_barrier(); /* g may alias out */
  if (lane < GQ_NVD) out[lane] = x;
  wave_barrier();
}

/* solve_tree_fused for TWO right-hand sides of the same tree-sparse system (out = S^-1 g, out2 = S^-1 g2): the elimination
 * of the matrix is shared, each extra right-hand side costs ~70 FMAs.  Used by the Sherman-Morrison step of an env with ONE
 * active row that couples two legs (newton_solve). */
__device__ inline void solve_tree_fused2(const float (*Sc)[9], const float (*Sb)[6], const float* g, float* out, const float* g2, float* out2) {
  const int lane = lane_id();
  { /* every lane runs the same instruction stream (lanes >= 4 mirror leg 0 and are discarded at the end): no branch, and the
     * cross-lane sums are called from wave-uniform control flow */
    const int hh = 3 * (lane < 4 ? lane : 0), t = hh + 1, c = hh + 2; /* joint indices of the leg (dof = 6 + joint) */
    float rc[9], rt[8], rh[7], bb[21], gb[6], hb[6];
#pragma unroll
    for (int j = 0; j < 6; j++) { rc[j] = Sc[c][j]; rt[j] = Sc[t][j]; rh[j] = Sc[hh][j]; }
    rc[6] = Sc[c][6]; rc[7] = Sc[c][7]; rc[8] = Sc[c][8];
    rt[6] = Sc[t][6]; rt[7] = Sc[t][7];
    rh[6] = Sc[hh][6];
    float gc = g[6 + c], gt = g[6 + t], gh = g[6 + hh];
    float hc = g2[6 + c], ht = g2[6 + t], hhh = g2[6 + hh];
    /* base block and right-hand side: lane 0 carries S_bb and g_b, the others start from zero (summed below) */
#pragma unroll
    for (int i = 0; i < 6; i++) {
      gb[i] = lane == 0 ? g[i] : 0.0f; hb[i] = lane == 0 ? g2[i] : 0.0f;
#pragma unroll
      for (int j = 0; j <= i; j++) bb[i * (i + 1) / 2 + j] = lane == 0 ? Sb[i][j] : 0.0f;
    }
    /* eliminate the calf: rows thigh, hip, base */
    const float ic = fast_rcp(rc[8]);
    {
      float f = rc[7] * ic;
#pragma unroll
      for (int j = 0; j <= 7; j++) rt[j] -= rc[j] * f;
      gt -= gc * f; ht -= hc * f;
      f = rc[6] * ic;
#pragma unroll
      for (int j = 0; j <= 6; j++) rh[j] -= rc[j] * f;
      gh -= gc * f; hhh -= hc * f;
#pragma unroll
      for (int i = 0; i < 6; i++) {
        f = rc[i] * ic;
#pragma unroll
        for (int j = 0; j <= i; j++) bb[i * (i + 1) / 2 + j] -= rc[j] * f;
        gb[i] -= gc * f; hb[i] -= hc * f;
      }
    }
    const float it = fast_rcp(rt[7]);
    {
      float f = rt[6] * it;
#pragma unroll
      for (int j = 0; j <= 6; j++) rh[j] -= rt[j] * f;
      gh -= gt * f; hhh -= ht * f;
#pragma unroll
      for (int i = 0; i < 6; i++) {
        f = rt[i] * it;
#pragma unroll
        for (int j = 0; j <= i; j++) bb[i * (i + 1) / 2 + j] -= rt[j] * f;
        gb[i] -= gt * f; hb[i] -= ht * f;
      }
    }
    const float ih = fast_rcp(rh[6]);
#pragma unroll
    for (int i = 0; i < 6; i++) {
      const float f = rh[i] * ih;
#pragma unroll
      for (int j = 0; j <= i; j++) bb[i * (i + 1) / 2 + j] -= rh[j] * f;
      gb[i] -= gh * f; hb[i] -= hhh * f;
    }
    /* Schur complement of the base: sum of the four legs */
#pragma unroll
    for (int q = 0; q < 21; q++) bb[q] = quad_sum(bb[q]);
#pragma unroll
    for (int i = 0; i < 6; i++) { gb[i] = quad_sum(gb[i]); hb[i] = quad_sum(hb[i]); }
    /* 6x6 base system, redundantly on each of the four lanes: elimination from the last row up (same order as the tree) */
    float xb[6], yb[6];
#pragma unroll
    for (int k = 5; k >= 0; k--) {
      const float inv = fast_rcp(bb[k * (k + 1) / 2 + k]);
#pragma unroll
      for (int i = 0; i < k; i++) {
        const float f = bb[k * (k + 1) / 2 + i] * inv;
#pragma unroll
        for (int j = 0; j <= i; j++) bb[i * (i + 1) / 2 + j] -= bb[k * (k + 1) / 2 + j] * f;
        gb[i] -= gb[k] * f; hb[i] -= hb[k] * f;
      }
      gb[k] *= inv; hb[k] *= inv; /* gb[k] now holds (g_k - sum_{j<k} S_kj x_j ... ) / S_kk once the x_j below are known */
      bb[k * (k + 1) / 2 + k] = inv;
    }
#pragma unroll
    for (int k = 0; k < 6; k++) {
      float s = gb[k], s2 = hb[k];
#pragma unroll
      for (int j = 0; j < k; j++) { const float cf = bb[k * (k + 1) / 2 + j] * bb[k * (k + 1) / 2 + k]; s -= cf * xb[j]; s2 -= cf * yb[j]; }
      xb[k] = s; yb[k] = s2;
    }
    /* back-substitution up the leg: hip, thigh, calf */
    float xh = gh, xt = gt, xc = gc, yh = hhh, yt = ht, yc = hc;
#pragma unroll
    for (int j = 0; j < 6; j++) { xh -= rh[j] * xb[j]; yh -= rh[j] * yb[j]; }
    xh *= ih; yh *= ih;
#pragma unroll
    for (int j = 0; j < 6; j++) { xt -= rt[j] * xb[j]; yt -= rt[j] * yb[j]; }
    xt = (xt - rt[6] * xh) * it; yt = (yt - rt[6] * yh) * it;
#pragma unroll
    for (int j = 0; j < 6; j++) { xc -= rc[j] * xb[j]; yc -= rc[j] * yb[j]; }
    xc = (xc - rc[6] * xh - rc[7] * xt) * ic; yc = (yc - rc[6] * yh - rc[7] * yt) * ic;
    wave_barrier(); /* g may alias out: every lane has read its right-hand side */
    if (lane < 4) { out[6 + hh] = xh; out[6 + t] = xt; out[6 + c] = xc; out2[6 + hh] = yh; out2[6 + t] = yt; out2[6 + c] = yc; }
    if (lane == 0) {
#pragma unroll
      for (int j = 0; j < 6; j++) { out[j] = xb[j]; out2[j] = yb[j]; }
    }
  }
  wave_barrier();
}

/* single right-hand side solve, leg-parallel: out <- (L'DL)^-1 g ; g, out: LDS [18] (may alias) */
__device__ inline void solve_tree_one(WaveMem& W, const float* F, const float* g, float* out) {
  const int lane = lane_id();
  float(*acc)[21] = W.acc2;
  float xc = 0.0f, xt = 0.0f, xh = 0.0f;
  const int hh = 6 + 3 * (lane & 3), t = hh + 1, c = hh + 2;
  if (lane < 4) { /* backward substitution inside the leg, contributions to the base collected per leg */
    xc = g[c];
    xt = g[t] - F[GQ_F_LC(c - 6, 7)] * xc;
    xh = g[hh] - F[GQ_F_LC(c - 6, 6)] * xc - F[GQ_F_LC(t - 6, 6)] * xt;
#pragma unroll
    for (int j = 0; j < 6; j++) acc[lane][j] = F[GQ_F_LC(c - 6, j)] * xc + F[GQ_F_LC(t - 6, j)] * xt + F[GQ_F_LC(hh - 6, j)] * xh;
  }
  wave_barrier();
  if (lane == 0) {
    float xb[6];
#pragma unroll
    for (int j = 0; j < 6; j++) xb[j] = g[j] - (acc[0][j] + acc[1][j] + acc[2][j] + acc[3][j]);
#pragma unroll
    for (int k = 5; k >= 1; k--)
#pragma unroll
      for (int j = 0; j < k; j++) xb[j] -= F[GQ_F_LB(k, j)] * xb[k];
#pragma unroll
    for (int k = 0; k < 6; k++) xb[k] *= F[GQ_F_DINV(k)];
#pragma unroll
    for (int k = 1; k < 6; k++)
#pragma unroll
      for (int j = 0; j < k; j++) xb[k] -= F[GQ_F_LB(k, j)] * xb[j];
#pragma unroll
    for (int k = 0; k < 6; k++) acc[4][k] = xb[k];
  }
  wave_barrier();
  if (lane < 4) {
    float xb[6];
#pragma unroll
    for (int j = 0; j < 6; j++) xb[j] = acc[4][j];
    xh *= F[GQ_F_DINV(hh)]; xt *= F[GQ_F_DINV(t)]; xc *= F[GQ_F_DINV(c)];
#pragma unroll
    for (int j = 0; j < 6; j++) { xh -= F[GQ_F_LC(hh - 6, j)] * xb[j]; xt -= F[GQ_F_LC(t - 6, j)] * xb[j]; xc -= F[GQ_F_LC(c - 6, j)] * xb[j]; }
    xt -= F[GQ_F_LC(t - 6, 6)] * xh;
    xc -= F[GQ_F_LC(c - 6, 6)] * xh + F[GQ_F_LC(c - 6, 7)] * xt;
    out[hh] = xh; out[t] = xt; out[c] = xc;
    if (lane == 0) {
#pragma unroll
      for (int j = 0; j < 6; j++) out[j] = xb[j];
    }
  }
  wave_barrier();
}

/* force law + cost of one row at residual y = J qacc - aref (mj_constraintUpdate); returns force, sets cost/active */
__device__ __forceinline__ float row_law(int rtype, float y, float R, float D, float floss, float& cost, float& wact) {
  float f = 0.0f;
  cost = 0.0f; wact = 0.0f;
  if (rtype == ROW_FRICTION) {
    if (y <= -R * floss) { f = floss; cost = -0.5f * R * floss * floss - floss * y; }
    else if (y >= R * floss) { f = -floss; cost = -0.5f * R * floss * floss + floss * y; }
    else { f = -D * y; cost = 0.5f * D * y * y; wact = D; }
  } else if (rtype != ROW_NONE) {
    if (y < 0.0f) { f = -D * y; cost = 0.5f * D * y * y; wact = D; }
  }
  return f;
}

/* which piece of its piecewise-quadratic cost a row is on at residual y */
__device__ __forceinline__ int row_piece(int rtype, float y, float R, float floss) {
  if (rtype == ROW_FRICTION) return y <= -R * floss ? 0 : (y >= R * floss ? 2 : 1);
  return (rtype != ROW_NONE && y < 0.0f) ? 1 : 0;
}

/* cost s_i(y) of one row at residual y (mj_constraintUpdate), branch-free: unilateral rows D min(y, 0)^2 / 2; friction-loss rows the
 * Huber function - quadratic inside |y| < R floss, linear outside */
__device__ __forceinline__ float row_cost(int rtype, float y, float R, float D, float floss) {
  const bool fr = rtype == ROW_FRICTION;
  const float lim = R * floss, ay = fabsf(y);
  const float quad = 0.5f * D * y * y;
  const float hub = ay < lim ? quad : floss * (ay - 0.5f * lim);
  return fr ? hub : ((rtype != ROW_NONE && y < 0.0f) ? quad : 0.0f);
}

/* derivative pieces of one row along the search direction (first and second derivative of s_i(y + alpha*v)) */
__device__ __forceinline__ void row_dd(int rtype, float y, float v, float R, float D, float floss, float& d1, float& d2) {
  /* selects, no branches: the line search evaluates this once per trial, and a lone tail wavefront pays 10-20 cycles for every divergent
   * branch (exec-mask save / restore around a handful of multiplies) - the trial was 150 instructions of which 20 were branches */
  const bool fr = rtype == ROW_FRICTION;
  const float lim = R * floss;
  const bool below = y <= -lim, above = y >= lim;
  const bool quad = fr ? !(below || above) : (rtype != ROW_NONE && y < 0.0f);
  const float lin = fr ? (below ? -floss * v : (above ? floss * v : 0.0f)) : 0.0f;
  d1 = quad ? D * y * v : lin; /* (association as mj_constraintUpdate's restatement in the oracle: bit-equal to the branchy form) */
  d2 = quad ? D * v * v : 0.0f;
}

/* Per-lane description of a row of an elliptic contact (CONE): code = e | dim << 4 (0: not such a row), r0 = lane of
 * the contact's normal row, fri = friction coefficient of this row (e >= 1), mu = friction_0 / sqrt(impratio),
 * D0 = 1 / R of the normal row.  Cost of the contact at residual z (mj_constraintUpdate): with N = mu z_0,
 * U_j = fri_j z_j, T = |U|:  top zone N >= mu T: 0;  bottom zone mu N + T <= 0: sum_j D_j z_j^2 / 2;  middle zone:
 * Dm (N - mu T)^2 / 2, Dm = D0 / (mu^2 (1 + mu^2)) - the dual of projecting onto the (scaled, circular) cone. */
struct EllRow { int code, r0; float fri, mu, D0; };

/* sum of `val` over the friction rows (e = 1 .. dim-1) of the lane's contact, available on every lane of the contact */
__device__ __forceinline__ float ell_seg_sum(const EllRow& E, float val) {
  const int dim = E.code >> 4;
  float s = 0.0f;
#pragma unroll
  for (int j = 1; j < 6; j++) {
    const float t = shfl_idx(val, E.r0 + j); /* wave-uniform call; lanes outside a contact read themselves */
    s += j < dim ? t : 0.0f;
  }
  return s;
}
__device__ __forceinline__ int ell_zone(float N, float T, float mu) { /* 0 top, 1 bottom, 2 middle */
  if (N >= mu * T || (T <= 0.0f && N >= 0.0f)) return 0;
  if (mu * N + T <= 0.0f || (T <= 0.0f && N < 0.0f)) return 1;
  return 2;
}

/* first and second derivative along the search direction of the lane's share of an elliptic contact's cost at step
 * alpha (lane-local: TT, UV, VV, y0, N1 are the contact sums at alpha = 0).  Bottom zone: every row its own quadratic;
 * middle zone: the normal row carries s = Dm q^2 / 2, q = N - mu T:  s' = Dm q q',  s'' = Dm (q'^2 + q q'') */
__device__ __forceinline__ void ell_dd(const EllRow& E, float alpha, float y, float v, float rD, float TT, float y0, float UV,
                                       float VV, float N1, float& d1, float& d2) {
  /* branch-free like row_dd (every zone's expression is evaluated, the lane's zone selects; quotients of a zone the lane is not in may be
   * inf / NaN and are discarded by the selects) */
  const float Na = E.mu * y0 + alpha * N1, TTa = fmaxf(0.0f, TT + 2.0f * alpha * UV + alpha * alpha * VV), Ta = fast_sqrt(TTa);
  const bool flat = Ta <= 0.0f;
  const bool top = Na >= E.mu * Ta || (flat && Na >= 0.0f);
  const bool bottom = !top && (E.mu * Na + Ta <= 0.0f || (flat && Na < 0.0f));
  const bool middle = !top && !bottom && (E.code & 15) == 0;
  const float Dm = fdiv(E.D0, E.mu * E.mu * (1.0f + E.mu * E.mu)), q = Na - E.mu * Ta;
  const float iT = fast_rcp(Ta), Tp = (UV + alpha * VV) * iT, qp = N1 - E.mu * Tp, Tpp = fmaxf(0.0f, VV - Tp * Tp) * iT;
  d1 = bottom ? rD * (y + alpha * v) * v : (middle ? Dm * q * qp : 0.0f);
  d2 = bottom ? rD * v * v : (middle ? Dm * (qp * qp - q * E.mu * Tpp) : 0.0f);
}

/* state of the lane's elliptic row at residual y (wave-uniform call): force, cost share (the normal row carries the
 * middle-zone cost of the contact), row weight for the Hessian (bottom zone only), zone, u_e = U_e / T, T^2 and z_0 */
__device__ __forceinline__ float ell_state(const EllRow& E, float y, float rD, float& ci, float& wact, int& zone, float& uhat,
                                           float& TT, float& y0) {
  const int e = E.code & 15;
  const float u = e >= 1 ? E.fri * y : 0.0f;
  TT = ell_seg_sum(E, u * u);
  y0 = shfl_idx(y, E.r0);
  const float N = E.mu * y0, T = fast_sqrt(TT);
  zone = ell_zone(N, T, E.mu);
  ci = 0.0f; wact = 0.0f; uhat = 0.0f;
  if (E.code == 0 || zone == 0) return 0.0f;
  if (zone == 1) { ci = 0.5f * rD * y * y; wact = rD; return -rD * y; }
  const float Dm = fdiv(E.D0, E.mu * E.mu * (1.0f + E.mu * E.mu)), q = N - E.mu * T;
  if (e == 0) { ci = 0.5f * Dm * q * q; return -Dm * q * E.mu; }
  uhat = fdiv(u, T);                   /* the middle zone's curvature is carried entirely by the virtual rows */
  return Dm * q * E.mu * E.fri * uhat;
}

/* Newton iterations.  In: row data in registers, smooth (= qfrc_smooth) / warm / Mc / Mb in LDS, J rows in W.u.B.
 * Out: W.qacc (solution), W.qfrc_c (= M (qacc - qacc_smooth) = J' f), returns the row's force; niter by reference. */
#ifndef GQ_HCHUNK
#define GQ_HCHUNK 4
#endif
/* relative tolerance of the line search on phi' (MuJoCo: opt.ls_tolerance = 0.01) */
/* relative step below which the iterate has converged to fp32 working precision (see the use site) */
#ifndef GQ_STEP_FLOOR
#define GQ_STEP_FLOOR 1e-6f
#endif
#ifndef GQ_LS_TRIALS
#define GQ_LS_TRIALS 16
#endif
#ifndef GQ_LS_HALVE
#define GQ_LS_HALVE 0.5f
#endif
#ifndef GQ_LS_WIDTH
#define GQ_LS_WIDTH 0.5f
#endif
#ifndef GQ_LS_TOL
#define GQ_LS_TOL 1e-2f
#endif
template <bool DBG, bool CONE>
__device__ inline float newton_solve(WaveMem& W, const StepConsts& m, const int fl_row_pre, const int (&hent_pre)[2], int rtype, float rR, float raref,
                                     float rfloss, int nefc, int nfl, int nsingle, int& niter, float* tdbg, const EllRow E, int prio,
                                     const bool xrow, const int xrow0) {
  const int lane = lane_id();
  /* the row's J lives in LDS (W.u.B[lane]) and is re-read where needed: 18 fewer registers across the iterations */
  /* (each use goes through opaque_ptr: otherwise the compiler merges the re-reads, keeps the 18 values live across the
   * whole solve and spills them to scratch - the opposite of the intent) */
#define JROW() (W.u.B[opaque_lane(lane)]) /* an opaque INDEX (an opaque pointer would lose its LDS address space and turn into flat
                                           * loads); taken ONCE per use site: each expansion re-derives lane * 72 with a slow v_mul_lo_u32 */
  const float rD = fast_rcp(rR);
  const float scale = m.nw_scale;
  float* dq = W.qacc_int;       /* scratch 18-vectors: free until S10 */
  float* Mdq = W.qfrc_c;
  float* grad = W.act;
  float* search = W.u2.n.nw[0];
  /* ---- starting point: the unconstrained acceleration qacc_smooth = M^-1 qfrc_smooth.  mj_fwdConstraint evaluates the
   * cost at qacc_warmstart and at qacc_smooth and starts from the cheaper one; the minimiser does not depend on the start
   * (strictly convex cost), only the iteration count does.  Measured on the benchmark's rollouts, qacc_smooth lies on the
   * solution's piece of the cost for 79 % of the envs (one Newton step, then the converged-step exit) while the previous
   * step's qacc does so for hardly any (torques change every step, the friction-loss rows follow them) - and the
   * comparison itself costs two residual evaluations, a mass-matrix product and two wave reductions.  So the warm start
   * is not consulted by this solver (it is still written, for PGS and for callers that read it). */
  solve_tree_fused<false, true>(W.Mc, W.Mb, W.F[0], 0.0f, W.smooth, W.qacc_smooth, GQ_EULER_FLEG(W), GQ_EULER_FBASE(W)); /* + the Euler system's factor, by the idle second quad */
  const int ld = lane < GQ_NVD ? lane : GQ_NVD - 1; /* mirror lanes (gq_step_body.h): lanes >= 18 repeat dof 17's reads, arithmetic and stores */
  W.qacc[ld] = W.qacc_smooth[ld];
  wave_barrier();
  float f = 0.0f;
  int iter = 0, exit_code = 0; /* why the loop ended (debug record, timer slot 23) */
  const int fl_row = fl_row_pre; /* (the dof's record: fetched at the start of the step, with the clamped lane) */
  /* the two Hessian entries this lane assembles every iteration: a host table (GqDevModel::newton_hent) - decoding the entry
   * index per lane cost ~80 instructions per step */
  int hent[2];
#pragma unroll
  for (int pass = 0; pass < 2; pass++) hent[pass] = hent_pre[pass]; /* (fetched in front of S5) */
  /* H without the elliptic virtual rows, kept in REGISTERS across the iterations (two entries per lane) together with the
   * weight every row currently has in it: an iteration adds  dw_r J_r' J_r  for the rows whose weight CHANGED - all active
   * rows in the first iteration (from M), the one to three rows that switched piece afterwards - instead of walking every
   * row for every entry every time. */
  float hbase[2], wprev = 0.0f;
#pragma unroll
  for (int pass = 0; pass < 2; pass++) { /* (every lane has two entries: the table's spare slots repeat entry 116; Mc | Mb are one flat array, slot = its index) */
    const int slot = (hent[pass] >> 16) & 0xff;
    hbase[pass] = (&W.Mc[0][0])[slot];
  }
  uint32_t tacc[7] = {0, 0, 0, 0, 0, 0, 0}, tprev = (DBG && tdbg) ? (uint32_t)cycles() : 0u; /* (32-bit: wave-uniform counters of the instrumented variant, half the registers) */
  if constexpr (DBG) if (tdbg && lane == 0) { tdbg[29] = 0.0f; tdbg[30] = 0.0f; tdbg[0] = 0.0f; } /* line-search trials, full-step shortcuts; slot 0 (no stage stamp uses it): largest count of active cross-leg rows in a dense step + 64 x dense steps + 4096 x Sherman-Morrison steps (slots 24 - 28 are the wave's clocks / hardware slot / priority, written at the start of step_wave) */
#define NW_T(i) do { if constexpr (DBG) if (tdbg) { const uint32_t tn = (uint32_t)cycles(); tacc[i] += tn - tprev; tprev = tn; } } while (0)
  NW_T(0);
  /* residual y = J qacc - aref and Ma-terms are evaluated once, then advanced incrementally along the search
   * direction (y += alpha J s, M dq += alpha M s), as mj_solNewton does */
  float y = -raref, md = 0.0f;
  {
    const float* J = JROW();
#pragma unroll
    for (int k = 0; k < GQ_NVD; k++) y += J[k] * W.qacc[k];
  }
  /* md = M (qacc - qacc_smooth) = M qacc - qfrc_smooth, advanced with the iterate; zero at the starting point */
  float gnorm2_prev = 0.0f, pred_prev = 1.0f;
  bool prev_unit = false; /* the previous step was a unit step taken because it lowered the cost (see the line search) */
  for (;; iter++) {
    /* a wave that needs many iterations decides when the launch ends: it moves ahead of the waves it shares the SIMD with */
    /* elliptic-cone models run 2.6 iterations on average and up to 14: with the pyramidal rule (priority 3 from the third iteration on) half of
     * the waves of a launch carried the top priority and it arbitrated nothing; their levels are spread over the iteration count instead
     * (3 / 5 / 7: go2 +1.8 %, hyqreal1 +2.5 %, spot +1.3 % - profiles/r04_priority_sweep.txt) */
    if constexpr (CONE) { const int want = iter >= 7 ? 3 : (iter >= 5 ? 2 : (iter >= 3 ? 1 : 0)); if (want > prio) { prio = want; wave_priority(prio); } }
    else if (iter + 1 > prio && iter > 0) { prio = iter + 1; wave_priority(prio); } /* (finer levels bought the pyramidal models nothing) */
    /* ---- constraint state at the current iterate */
    float ci, wact;
    f = row_law(rtype, y, rR, rD, rfloss, ci, wact);
    int zone = 0;
    float uhat = 0.0f, TT = 0.0f, y0 = 0.0f;
    if constexpr (CONE) {
      float c2, w2;
      const float f2 = ell_state(E, y, rD, c2, w2, zone, uhat, TT, y0);
      if (E.code) { f = f2; ci = c2; wact = w2; }
    }
    W.force[lane] = f; /* row forces, read column-wise for J'f below */
    Mdq[ld] = md;
    wave_barrier();
    /* (MuJoCo's improvement test compares successive costs; in fp32 their round-off - 1e-7 of a cost dominated by stiff
     * contact rows - is far above `tolerance`, so the test is applied to the decrease predicted by the line search,
     * after the step, below) */
    if (iter >= m.iterations) { exit_code = 2; break; }
    NW_T(1);
    /* ---- gradient = M dq - J' f  (lane = dof walks its column of J in LDS) */
    float gd = 0.0f, gterm = 0.0f;
    { /* lane = dof, mirror lanes (masked in the sums below) */
      /* friction-loss rows are e_dof: their force lands on one dof, only limit / contact rows are walked */
      /* rows in chunks of four, all eight LDS reads of a chunk in flight together (one LDS latency per chunk instead of
       * one per row); rows past nefc hold zero forces (lanes >= nefc write 0) and row indices stay below 64 */
      float s0 = fl_row >= 0 ? W.force[fl_row] : 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
      if constexpr (!CONE) { /* pyramidal models: mostly 0-8 contact rows, the two-row walk is cheaper there */
        int r = nfl;
        for (; r + 2 <= nefc; r += 2) { s0 += W.u.B[r][ld] * W.force[r]; s1 += W.u.B[r + 1][ld] * W.force[r + 1]; }
        if (r < nefc) s0 += W.u.B[r][ld] * W.force[r];
      } else
      for (int r = nfl; r < nefc; r += 4) {
        const int r1 = r + 1 < 64 ? r + 1 : 63, r2 = r + 2 < 64 ? r + 2 : 63, r3 = r + 3 < 64 ? r + 3 : 63;
        const float b0 = W.u.B[r][ld], b1 = W.u.B[r1][ld], b2 = W.u.B[r2][ld], b3 = W.u.B[r3][ld];
        const float f0 = W.force[r], f1 = r + 1 < nefc ? W.force[r1] : 0.0f, f2 = r + 2 < nefc ? W.force[r2] : 0.0f, f3 = r + 3 < nefc ? W.force[r3] : 0.0f;
        s0 += b0 * f0; s1 += b1 * f1; s2 += b2 * f2; s3 += b3 * f3;
      }
      s0 += s2; s1 += s3;
      gd = md - (s0 + s1);
      gterm = md * md + (s0 + s1) * (s0 + s1);
    }
    const float gnorm2 = wave_sum(lane < GQ_NVD ? gd * gd : 0.0f);
    if (scale * fast_sqrt(gnorm2) < m.tolerance) { exit_code = 3; break; }
    /* fp32 floor (GqModelDesc.noise_floor): the gradient is a difference of two vectors; once it is down at their
     * round-off a further Newton step only chases noise */
    /* (not straight after a unit step taken on its cost decrease: rows changed piece there, and the iterate it leaves has not been through
     * a step on its own pieces yet - a hyqreal2 pose lying on the floor reached this ratio three such steps into an eight-iteration solve) */
    if (iter > 0 && !prev_unit && m.noise_floor > 0.0f && gnorm2 <= m.noise_floor * m.noise_floor * wave_sum(lane < GQ_NVD ? gterm : 0.0f)) { exit_code = 4; break; }
    /* stagnation at working precision: close to the solution (the last step promised less than 1e-6, scaled like
     * `tolerance`) Newton's gradient collapses from one iterate to the next; one that did not even halve is rounding noise
     * of the stiff rows' residuals (elliptic models, impratio 100: the iterates then cycle between neighbouring fp32
     * states, every step still "promising" a few 1e-8 - above `tolerance` - until the iteration cap). */
    if constexpr (CONE) { /* (pyramidal problems are piecewise quadratic: the converged-step test below ends them exactly) */
      if (iter > 0 && pred_prev < 1e-6f && gnorm2 >= 0.25f * gnorm2_prev) { exit_code = 8; break; }
      gnorm2_prev = gnorm2;
    }
    grad[ld] = -gd; /* right-hand side of H search = -grad */
    wave_barrier();
    /* a contact between two different legs couples them in H = M + J'DJ, which then no longer has M's tree sparsity - but
     * only while one of its rows is active (wave-uniform tests; robot-robot rows are the last ones, + their virtual rows).
     * ONE active coupling row c (the common case with frictionless leg geoms): H = H0 + w c c' with H0 tree-sparse, solved
     * by Sherman-Morrison on two tree solves that share their elimination.  More: the dense lane-parallel step. */
    bool xl = false, xsm = false;
    int xr = 0;
    if (xrow0 >= 0) { /* -1: no contact couples two legs (0 is a valid first row: models without friction loss) */
      const uint64_t xm = ballot(xrow && (wact != 0.0f || (CONE && zone == 2)));
      xl = xm != 0;
      xsm = !CONE && popc64(xm) == 1;
      xr = xsm ? ffs64(xm) : 0;
    }
    if constexpr (DBG) if (tdbg && xl) { /* wave-uniform */
      const float kx = (float)popc64(ballot(xrow && (wact != 0.0f || (CONE && zone == 2))));
      if (lane == 0) {
        const int w0 = (int)tdbg[0];
        if (xsm) tdbg[0] = (float)(w0 + 4096);
        else tdbg[0] = (float)((w0 & ~63) + 64 + imax(w0 & 63, imin((int)kx, 63)));
      }
    }
    const float xw = xsm ? bcast(wact, xr) : 0.0f;
    /* Hessian weight of the row (the Sherman-Morrison row stays out of the tree-sparse part) and its change since the last
     * assembly; the CHANGES replace the forces in W.force */
    const float wh = (xsm && lane == xr) ? 0.0f : wact;
    const float dw = lane < nefc ? wh - wprev : 0.0f;
    wprev = wh;
    const uint64_t chg = ballot(dw != 0.0f);
    W.force[lane] = dw;
    wave_barrier();
    int nrowh = nefc; /* rows the Hessian assembly walks */
    if constexpr (CONE) {
      /* a contact in the middle zone has the dense block  Dm g g' + Dm kappa F^(1/2) (I - u u') F^(1/2)  (F = diag of the
       * squared friction coefficients, u = U / T, kappa = -mu q / T > 0).  It is added as virtual rows above nefc (S6
       * reserved the space), every one with a POSITIVE weight so that H stays positive semi-definite in fp32 whatever the
       * round-off:  a = sum_i g_i J_i (weight Dm)  and, for an orthonormal basis p_k of the complement of u,
       * b_k = sum_j fri_j p_kj J_j (weight Dm kappa).  dim 3: p = (-u_2, u_1);  dim 6: columns 2..5 of the Householder
       * reflection that maps e_1 to -+u.  (Formed as diag - w w' the projector loses definiteness by ~1e-7 Dm kappa, which
       * is comparable to the inertia of a light leg.) */
      /* Built lane-parallel (a wave-uniform loop over the contacts with v_readlane broadcasts was 7k cycles per iteration for
       * a wave alone on its SIMD).  Every virtual row is a combination  sum_i coef_i J[r0 + i]  of its contact's rows:
       * (1) the head lanes of the middle-zone contacts count the virtual rows (ballots, no scan: a contact has 2 or 5);
       * (2) row lane r0 + i writes ITS coefficient of each of its contact's virtual rows into the (still unused) storage of
       *     that virtual row, the head lane adds r0, dim and the weight;
       * (3) the nvirt x 18 outputs are spread over all 64 lanes (registers), and written back after a barrier. */
      int ecode = E.code, er0 = E.r0;
      opaque(ecode); opaque(er0); /* as for the Hessian indices below: keep the derived lane indices / addresses out of the loop's live set */
      const int e = ecode & 15, dimc = ecode >> 4;
      const bool mid = ecode != 0 && zone == 2, head = mid && e == 0;
      const uint64_t m3 = ballot(head && dimc == 3), m6 = ballot(head && dimc == 6);
      if ((m3 | m6) != 0) { /* wave-uniform */
        const uint64_t below = (1ull << (er0 & 63)) - 1ull;
        const int vb = nefc + 2 * popc64(m3 & below) + 5 * popc64(m6 & below); /* first virtual row of the lane's contact */
        const int nvirt = 2 * popc64(m3) + 5 * popc64(m6);
        const int mdim = m6 ? 6 : 3;
        float uh[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}; /* u_j of the lane's contact ([0] unused) */
#pragma unroll
        for (int j = 1; j < 6; j++)
          if (j < mdim) { const float t = shfl_idx(uhat, er0 + j); uh[j] = j < dimc ? t : 0.0f; }
        if (mid) {
          const float Tc = fast_sqrt(TT), qc = E.mu * y0 - E.mu * Tc;
          const float Dm = fdiv(E.D0, E.mu * E.mu * (1.0f + E.mu * E.mu)), wk = fdiv(Dm * (-E.mu * qc), Tc);
          W.u.B[vb][e] = e == 0 ? E.mu : -E.mu * E.fri * uhat;
          if (dimc == 3) {
            W.u.B[vb + 1][e] = e == 0 ? 0.0f : (e == 1 ? -uh[2] * E.fri : uh[1] * E.fri);
          } else {
            const float sg = uh[1] < 0.0f ? -1.0f : 1.0f, inv = fast_rcp(1.0f + fabsf(uh[1])); /* 2 / (h'h), h'h = 2 (1 + |u_1|) */
            const float hve = e == 0 ? 0.0f : uhat + (e == 1 ? sg : 0.0f);
#pragma unroll
            for (int k = 2; k < 6; k++) W.u.B[vb + k - 1][e] = e == 0 ? 0.0f : ((e == k ? 1.0f : 0.0f) - hve * uh[k] * inv) * E.fri;
          }
          if (e == 0) {
            for (int k = 0; k < dimc - 1; k++) {
              W.u.B[vb + k][6] = __builtin_bit_cast(float, er0 | (dimc << 8));
              W.force[vb + k] = k == 0 ? Dm : wk;
            }
          }
        }
        wave_barrier();
        const int total = nvirt * GQ_NVD; /* <= 25 virtual rows (five condim-6 contacts in the middle zone) x 18 = 450 outputs: 8 per lane */
        float outv[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
          outv[k] = 0.0f;
          if (64 * k < total) { /* wave-uniform */
            const int o = lane + 64 * k < total ? lane + 64 * k : total - 1;
            const int v = (o * 3641) >> 16, d = o - GQ_NVD * v; /* o / 18 for o < 1024 */
            const float* Tv = W.u.B[nefc + v];
            const int meta = __builtin_bit_cast(int, Tv[6]), r0 = meta & 0xff, dv = meta >> 8;
            float acc = 0.0f;
#pragma unroll
            for (int i = 0; i < 6; i++)
              if (i < mdim) { /* wave-uniform */
                const int ri = r0 + i < 64 ? r0 + i : 63;
                const float c = Tv[i], Jv = W.u.B[ri][d];
                acc += i < dv ? c * Jv : 0.0f;
              }
            outv[k] = acc;
          }
        }
        wave_barrier();
#pragma unroll
        for (int k = 0; k < 8; k++)
          if (64 * k < total && lane + 64 * k < total) {
            const int o = lane + 64 * k, v = (o * 3641) >> 16, d = o - GQ_NVD * v;
            W.u.B[nefc + v][d] = outv[k];
          }
        nrowh = nefc + nvirt;
      }
      wave_barrier();
    }
    NW_T(2);
    /* ---- Hessian in M's tree-sparse layout: H = M + sum_r w_r J_r' J_r.  117 structurally non-zero entries, two per
     * lane.  Friction-loss rows are +-e_dof: their weight change goes straight to the diagonal; the changed limit / contact
     * rows are walked wave-uniformly (row index and weight change in SGPRs, the two J entries of each of the lane's
     * entries from LDS), two rows per trip so that their reads are in flight together. */
    {
      /* (elliptic variants: the decoded indices and LDS addresses below are loop invariants that LLVM hoists out of the Newton
       * loop and then spills - 17 scratch reloads per iteration; decoding them again from an opaque copy is 12 VALU) */
      int h0 = hent[0], h1 = hent[1];
      if constexpr (CONE) { opaque(h0); opaque(h1); }
      const int e0 = h0, e1 = h1;
      const int da0 = e0 & 0xff, db0 = (e0 >> 8) & 0xff, da1 = e1 & 0xff, db1 = (e1 >> 8) & 0xff;
      const uint64_t flm = nfl >= 64 ? ~0ull : ((1ull << nfl) - 1ull);
      if (chg & flm) {
        const int f0 = (e0 >> 24) - 1, f1 = (e1 >> 24) - 1;
        if (f0 >= 0) hbase[0] += W.force[f0];
        if (f1 >= 0) hbase[1] += W.force[f1];
      }
      uint64_t mk = chg & ~flm;
      while (mk) {
        const int ra = ffs64(mk);
        mk &= mk - 1;
        const bool two = mk != 0;
        const int rb = two ? ffs64(mk) : ra;
        mk &= mk - 1; /* 0 stays 0 */
        const float wa = bcast(dw, ra), wb = two ? bcast(dw, rb) : 0.0f;
        const float a00 = W.u.B[ra][da0], a01 = W.u.B[ra][db0], a10 = W.u.B[ra][da1], a11 = W.u.B[ra][db1];
        const float b00 = W.u.B[rb][da0], b01 = W.u.B[rb][db0], b10 = W.u.B[rb][da1], b11 = W.u.B[rb][db1];
        hbase[0] += wa * a00 * a01 + wb * b00 * b01;
        hbase[1] += wa * a10 * a11 + wb * b10 * b11;
      }
      float hv0 = hbase[0], hv1 = hbase[1];
      if constexpr (CONE) { /* the virtual rows of the middle-zone contacts are rebuilt every iteration */
        float s2 = 0.0f, s3 = 0.0f;
        for (int r = nefc; r < nrowh; r += 2) {
          const int r1 = r + 1 < 64 ? r + 1 : 63;
          const float w0 = W.force[r], w1 = r + 1 < nrowh ? W.force[r1] : 0.0f;
          const float a00 = W.u.B[r][da0], a01 = W.u.B[r][db0], a10 = W.u.B[r][da1], a11 = W.u.B[r][db1];
          const float b00 = W.u.B[r1][da0], b01 = W.u.B[r1][db0], b10 = W.u.B[r1][da1], b11 = W.u.B[r1][db1];
          hv0 += w0 * a00 * a01; s2 += w1 * b00 * b01;
          hv1 += w0 * a10 * a11; s3 += w1 * b10 * b11;
        }
        hv0 += s2; hv1 += s3;
      }
#pragma unroll
      for (int pass = 0; pass < 2; pass++) { /* Hc | Hb are one flat array like Mc | Mb; a base entry is stored on both sides of the diagonal, a leg entry twice in place */
        const int hp = pass ? h1 : h0;
        const int da = hp & 0xff, db = (hp >> 8) & 0xff, slot = (hp >> 16) & 0xff;
        const float hv = pass ? hv1 : hv0;
        float* H0 = &W.u2.n.Hc[0][0];
        H0[slot] = hv;
        H0[slot < 108 ? slot : 108 + 6 * db + da] = hv;
      }
      if (xl && !xsm) { /* the dense step reads the coupling rows' WEIGHTS from W.force (virtual rows above nefc hold theirs) */
        wave_barrier();
        if (lane < nefc) W.force[lane] = wh;
      }
    }
    wave_barrier();
    NW_T(3);
    if (xsm) {
      float* z1 = W.u2.n.nw[1];
      solve_tree_fused2(W.u2.n.Hc, W.u2.n.Hb, grad, search, W.u.B[xr], z1);
      const float cr = lane < GQ_NVD ? W.u.B[xr][lane] : 0.0f;
      const float cz0 = wave_sum(lane < GQ_NVD ? cr * search[lane] : 0.0f), cz1 = wave_sum(lane < GQ_NVD ? cr * z1[lane] : 0.0f);
      const float lam = fdiv(cz0, fast_rcp(xw) + cz1);
      wave_barrier();
      if (lane < GQ_NVD) search[lane] -= lam * z1[lane];
      wave_barrier();
    } else if (xl) newton_dense_step(W, xrow0, nrowh, grad, search);
    else solve_tree_fused<false>(W.u2.n.Hc, W.u2.n.Hb, nullptr, 0.0f, grad, search);
    NW_T(4);
    /* ---- exact line search on phi(alpha) = cost(qacc + alpha search): safeguarded Newton on phi' */
    float v = 0.0f;
    {
      const float* J = JROW();
#pragma unroll
      for (int k = 0; k < GQ_NVD; k++) v += J[k] * search[k];
    }
    /* (elliptic variants: the row's LDS addresses are derived again from an opaque lane index every iteration - hoisted out of the loop
     * they were kept in scratch memory and reloaded here, four loads and their wait per iteration) */
    const float ms = mul_m_row(W, search, CONE ? opaque_lane(ld) : ld);
    float alpha = 0.0f, lo = 0.0f, hi = -1.0f; /* hi < 0: no upper bracket yet */
    bool first_try = false;
    float g0 = 0.0f, UV = 0.0f, VV = 0.0f, N1 = 0.0f;
    /* the cost is piecewise quadratic in every row's residual, the pieces being intervals: if the full Newton step leaves
     * every row on the piece the Hessian was assembled for, phi is one parabola on [0, 1] and its minimiser is the Newton
     * step itself - no derivative sums, no trial steps (the usual last iteration of a solve) */
    bool full_step = false;
    if constexpr (!CONE) full_step = ballot(row_piece(rtype, y, rR, rfloss) != row_piece(rtype, y + v, rR, rfloss)) == 0;
    NW_T(5);
    if constexpr (DBG) if (tdbg && lane == 0 && full_step) tdbg[30] += 1.0f;
    /* phi'(alpha) = sum over lanes of (s.Mdq + alpha s.Ms) [dof lanes] + d1(alpha) [row lanes]: the quadratic part rides
     * in the same reduction as the rows' derivatives */
    const float p1 = lane < GQ_NVD ? search[ld] * md : 0.0f, p2 = lane < GQ_NVD ? search[ld] * ms : 0.0f;
    /* The unit step is taken whenever it LOWERS the cost (one reduction: phi(1) - phi(0) = s.Mdq + s.Ms / 2 + the rows' cost changes),
     * without a search for the minimiser along the direction.  The minimiser of the strictly convex problem does not depend on the step rule,
     * only the path to it does: over 20 000 benchmark-like steps of the fp64 restatement (tools/newton_start_experiment.py, GQO_LS_MODE=1)
     * the unit step lowers the cost in 96.8 % of the iterations and the iteration count FALLS (mean 1.346 -> 1.321, three or more iterations
     * 1 623 -> 1 159 of 20 000) against the exact line search - across the kink of a row that changes piece, the exact minimiser along the
     * direction is no better a place to linearise again than the Newton point.  What it removes is the search itself: two wave reductions
     * and five branches per trial, 5 - 22 trials in the waves that end a launch (profiles/r05_lone_wave_stages256.txt).  When the unit step
     * does not lower the cost (3 % of the iterations) the safeguarded search below runs as before. */
    float dphi = 0.0f; /* phi(1) - phi(0) when the unit step was taken on that evidence */
    bool unit_step = false;
    /* (elliptic models keep the search.  The rule holds for any convex cost - the unit step lowers it in 89 % / 82 % of the iterations of go2 /
     * hyqreal1 in the fp64 restatement - and it was tried twice: as here (go2 26.0 -> 8.6 M: fp32 noise in the sign of small cost differences of
     * the stiff cone rows, and the extra registers spilled), and as an Armijo test phi(1) - phi(0) <= 1e-4 phi'(0) riding as a fourth reduction
     * in the search's first pass (mean solver time -8 %, but the iteration histogram grows a tail - 16 instead of 12 iterations at the end,
     * three times as many envs at 8 - 11 - and the launch lasts as long as its slowest wave: go2 27.2 -> 24.0 M, hyqreal1 31.1 -> 29.1 M)) */
    if constexpr (!CONE) if (!full_step) { /* wave-uniform */
      const float dc = row_cost(rtype, y + v, rR, rD, rfloss) - row_cost(rtype, y, rR, rD, rfloss);
      dphi = wave_sum(p1 + 0.5f * p2 + dc);
      unit_step = dphi < 0.0f;
      if (unit_step && scale * (-dphi) < m.tolerance) { /* wave-uniform */
        /* a decrease below the tolerance ENDS the solve (small_step below): it must be the decrease of a converged point, not of a unit step that
         * overshot across a kink and landed at nearly the same cost.  The line's slope tells them apart - at a minimiser phi'(0) = g.s is itself
         * below the tolerance (-phi'(0) / 2 is the decrease a quadratic would give); otherwise the safeguarded search runs */
        float d1c, d2c;
        row_dd(rtype, y, v, rR, rD, rfloss, d1c, d2c);
        unit_step = scale * (-0.5f * wave_sum(p1 + d1c)) < m.tolerance;
      }
      if constexpr (DBG) if (tdbg && lane == 0 && unit_step) tdbg[30] += 1.0f;
    }
    if (full_step) { alpha = 1.0f; first_try = true; }
    else if (unit_step) alpha = 1.0f;
    else {
    float d1, d2;
    row_dd(rtype, y, v, rR, rD, rfloss, d1, d2);
    /* elliptic contacts: along the line T(alpha)^2 = TT + 2 alpha UV + alpha^2 VV and N(alpha) = N + alpha N1, so three
     * contact sums taken once make every trial step a lane-local evaluation */
    if constexpr (CONE) {
      const bool fr = (E.code & 15) >= 1;
      const float u = fr ? E.fri * y : 0.0f, V = fr ? E.fri * v : 0.0f;
      UV = ell_seg_sum(E, u * V); VV = ell_seg_sum(E, V * V); N1 = E.mu * shfl_idx(v, E.r0);
      float e1, e2;
      ell_dd(E, 0.0f, y, v, rD, TT, y0, UV, VV, N1, e1, e2);
      d1 = E.code ? e1 : d1; d2 = E.code ? e2 : d2;
    }
    /* The search direction solves H s = -grad with the Hessian of the pieces the rows are on, so phi'(0) / phi''(0) = -1: the first trial is
     * the unit step, and its two sums ride with phi'(0) in ONE pass of three interleaved reductions instead of waiting for phi'(0) and
     * phi''(0) first (go2 +0.5 %, hyqreal1 +1.5 %, mini_cheetah +0.4 %) */
    float d1u, d2u;
    row_dd(rtype, y + v, v, rR, rD, rfloss, d1u, d2u);
    if constexpr (CONE) {
      float e1, e2;
      ell_dd(E, 1.0f, y, v, rD, TT, y0, UV, VV, N1, e1, e2);
      d1u = E.code ? e1 : d1u; d2u = E.code ? e2 : d2u;
    }
    g0 = wave_sum(p1 + d1);
    const float ga_u = wave_sum(p1 + p2 + d1u), ha_u = wave_sum(p2 + d2u);
    if (!(g0 < 0.0f)) { exit_code = 5; break; } /* not a descent direction: converged to working precision */
    alpha = 1.0f;
    /* Newton on phi' while it makes progress; phi' is only piecewise smooth (a row changing piece, an elliptic contact whose
     * tangential residual passes near zero), and across a kink whose slopes differ by more than 2x Newton steps from the
     * two sides overshoot each other for ever inside the bracket.  So once a bracket exists, a trial that did not halve
     * |phi'| is followed by the bracket's midpoint.  If the trials run out, the step is the bracket's lower end: phi' < 0
     * there, so the cost did go down (the unverified last candidate could sit beyond the minimiser far enough to RAISE the
     * cost - spot cycled on that until the iteration cap, with a wrong qacc). */
    float gprev = fabsf(g0);
    bool done = false;
    float wd1 = 1e30f, wd2 = 1e30f; /* bracket widths after the last two trials (pyramidal rule) */
    float ga = ga_u, ha = ha_u;
    for (int ls = 0;; ls++) {
      if constexpr (DBG) if (tdbg && lane == 0) tdbg[29] += 1.0f;
      if (fabsf(ga) <= GQ_LS_TOL * fabsf(g0)) { first_try = ls == 0; done = true; break; } /* an approximate line search, like MuJoCo's */
      const bool neg = ga < 0.0f;
      lo = neg ? alpha : lo; hi = neg ? hi : alpha;
      const float newton = alpha - fdiv(ga, ha);
      const bool bracketed = hi > 0.0f;
      const bool outside = !(newton > lo) || (bracketed && !(newton < hi));
      bool bisect;
      if constexpr (CONE) bisect = outside || fabsf(ga) > GQ_LS_HALVE * gprev;
      else { const float wd = hi - lo; bisect = outside || wd > GQ_LS_WIDTH * wd2; wd2 = bracketed ? wd1 : wd2; wd1 = bracketed ? wd : wd1; }
      const float an = bracketed ? (bisect ? 0.5f * (lo + hi) : newton) : (outside ? 2.0f * alpha : newton);
      gprev = fabsf(ga);
      alpha = an;
      if (ls + 1 >= GQ_LS_TRIALS) break;
      row_dd(rtype, y + alpha * v, v, rR, rD, rfloss, d1, d2);
      if constexpr (CONE) {
        float e1, e2;
        ell_dd(E, alpha, y, v, rD, TT, y0, UV, VV, N1, e1, e2);
        d1 = E.code ? e1 : d1; d2 = E.code ? e2 : d2;
      }
      ga = wave_sum(p1 + alpha * p2 + d1);
      ha = wave_sum(p2 + d2);
    }
    if (!done) alpha = lo > 0.0f ? lo : alpha; /* lo == 0: every trial overshot; the last midpoint is the best guess */
    }
    wave_barrier();
#ifdef GQ_EMU_TRACE /* host emulator only (tests/simt_emu, -DGQ_EMU_TRACE): one line per Newton iteration */
    if (lane == 0 && getenv("GQ_EMU_TRACE")) printf("it %d gnorm %.6e g0 %.6e alpha %.6e first %d scale*pred %.3e  lo %.3e hi %.3e\n", iter, (double)sqrtf(gnorm2), (double)g0, (double)alpha, (int)first_try, (double)(scale * (-0.5f * g0 * alpha)), (double)lo, (double)hi);
#endif
    /* fp32 resolution of the iterate: a step that moves no component by more than GQ_STEP_FLOOR (relative, 1 for small
     * components - the metric of the parity tests) is the last one.  Newton converges quadratically, so what remains
     * after such a step is far below it; without this test a stiff elliptic problem (impratio 100) can sit at a gradient
     * of 1e-7 of its starting value - rounding noise of the residuals, yet above `tolerance` and above the noise floor
     * of the gradient's two terms - and repeat a step that no longer changes qacc until the iteration cap (go1: one env
     * in 10 000 env-steps, 100 iterations, 1.1 ms for the whole launch). */
    const float qa_old = W.qacc[ld], qa_new = qa_old + alpha * search[ld];
    const bool tiny_step = CONE && ballot(lane < GQ_NVD && fabsf(alpha * search[ld]) > GQ_STEP_FLOOR * fmaxf(1.0f, fabsf(qa_old))) == 0;
    if constexpr (!CONE) wave_barrier(); /* (the ballot is one for the emulator: read above, write below) */
    W.qacc[ld] = qa_new; md += alpha * ms;
    const float ynew = y + alpha * v;
    /* the cost is piecewise quadratic.  A full Newton step (accepted at the first trial) that leaves every row on the
     * piece it was linearised on has reached the minimiser of a model that IS the cost there: converged, and the
     * usual extra iteration that only re-evaluates cost and gradient to find that out is skipped */
    bool moved = row_piece(rtype, y, rR, rfloss) != row_piece(rtype, ynew, rR, rfloss);
    if constexpr (CONE) if (E.code) { /* the middle zone is not quadratic: a contact there (before or after) always counts as moved */
      const float TTn = fmaxf(0.0f, TT + 2.0f * alpha * UV + alpha * alpha * VV);
      const int zn = ell_zone(E.mu * y0 + alpha * N1, fast_sqrt(TTn), E.mu);
      moved = zn != zone || zone == 2;
    }
    y = ynew;
    wave_barrier();
    NW_T(6);
    /* improvement of this step from the line-search model (exact for a quadratic phi): phi(0) - phi(alpha) = -g0 alpha / 2 */
    /* (a unit step taken on its cost decrease: the decrease itself - MuJoCo's own `improvement`) */
    const float pred = unit_step ? scale * (-dphi) : scale * (-0.5f * g0 * alpha);
    prev_unit = unit_step;
    if constexpr (CONE) pred_prev = pred;
    const bool small_step = pred < m.tolerance || tiny_step;
    if ((first_try && ballot(moved) == 0) || small_step) {
      float ci, wact;
      f = row_law(rtype, y, rR, rD, rfloss, ci, wact);
      if constexpr (CONE) {
        int z2; float t1, t2, t3;
        const float f2 = ell_state(E, y, rD, ci, wact, z2, t1, t2, t3);
        if (E.code) f = f2;
      }
      Mdq[ld] = md;
      iter++;
      exit_code = tiny_step ? 7 : (small_step ? 1 : 6);
      break;
    }
  }
  if constexpr (DBG) if (tdbg && lane == 0)
  { for (int k = 0; k < 7; k++) tdbg[16 + k] = (float)tacc[k]; tdbg[23] = (float)exit_code; }
#undef NW_T
#undef JROW
  niter = iter;
  wave_barrier();
  return f;
}

}  // namespace gq
