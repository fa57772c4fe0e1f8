/*
 * gq_boxes.h - narrow phase against the static world boxes of a scene (terrain.py add_box :121-142: random_boxes,
 * random_pyramids, ramp, slippery, stairs), BOXES variants of the step kernel only (Newton solver).
 *
 * Per env and step: lane = box picks the boxes whose bounding sphere meets the robot's (ballot), then for every such
 * box the collision items (4 foot spheres + link vertex clouds) are tested: a sphere against a box is exact (centre
 * clamped into the box, or pushed out through the nearest face when inside - MuJoCo's mjc_SphereBox); a link geom is
 * its deepest cloud vertex inflated by the cloud radius, found by a 64-lane scan.  That is NOT MuJoCo's mesh-box routine
 * (libccd penetration of the hulls) but coincides with it for vertex-on-face touching; restated in the oracle
 * (gqo_collision) the same way.  Contacts are appended to the floor's in (box, geom id) order with their normal;
 * tangents follow mju_makeFrame.
 *
 * Everything is expressed relative to the env's base x/y like the rest of the kernel (box position minus base x/y is
 * taken in f64 first).  Scratch lives in the L'DL factor block of WaveMem, which the Newton path does not use.
 */
#pragma once
#include "gq_step_kernel.h"

namespace gq {

/* LDS scratch carved out of WaveMem::F (Newton path: F[0][0..17] = h*damping, the rest is free) */
#define GQ_BX_WCLS(W) (reinterpret_cast<int32_t*>(&(W).F[0][32]))   /* [12] world geom of contact c: -1 floor, else box class */
#define GQ_BX_LGNRM(W) (&(W).F[1][0])                                  /* [GQ_MAXLG][3] normal of the geom's hit on the current box */
#define GQ_BX_CONNRM(W) (&(W).F[1][3 * GQ_MAXLG])                      /* [12][3] contact normals */

/* sphere of radius r centred at c (box frame) against a box of half extents s: signed distance, outward normal n (box frame) */
__device__ __forceinline__ float sphere_box(V3 c, V3 s, float r, V3& n) {
  const V3 q = v3(med3(c.x, -s.x, s.x), med3(c.y, -s.y, s.y), med3(c.z, -s.z, s.z));
  const V3 d = c - q;
  const float l2 = dot(d, d);
  if (l2 > 0.0f) {
    const float inv = fast_rsqrt(l2);
    n = inv * d;
    return l2 * inv - r;
  }
  /* centre inside the box: leave through the nearest face */
  const float ex = s.x - fabsf(c.x), ey = s.y - fabsf(c.y), ez = s.z - fabsf(c.z);
  if (ex <= ey && ex <= ez) { n = v3(c.x >= 0.0f ? 1.0f : -1.0f, 0.0f, 0.0f); return -ex - r; }
  if (ey <= ez) { n = v3(0.0f, c.y >= 0.0f ? 1.0f : -1.0f, 0.0f); return -ey - r; }
  n = v3(0.0f, 0.0f, c.z >= 0.0f ? 1.0f : -1.0f);
  return -ez - r;
}

/* mju_makeFrame: tangents of a contact frame from its normal */
__device__ __forceinline__ void make_frame(V3 n, V3& t1, V3& t2) {
  V3 y = (n.y < 0.5f && n.y > -0.5f) ? v3(0.0f, 1.0f, 0.0f) : v3(0.0f, 0.0f, 1.0f);
  const float d = dot(n, y);
  y = y - d * n;
  t1 = fast_rsqrt(dot(y, y)) * y;
  t2 = cross(n, t1);
}

/* bit mask (two words) of the boxes whose bounding sphere meets the robot's; base = (0, 0, basez) in kernel coordinates */
__device__ inline void box_candidates(const WaveMem& W, const GQ_MODEL GqDevModel& m, double bx, double by, float zoff, uint64_t cand[2]) {
  const int lane = lane_id();
#pragma unroll
  for (int half = 0; half < 2; half++) {
    const int b = half * GQ_WAVE + lane;
    bool near = false;
    if (b < m.nbox) {
      const GQ_MODEL GqDevBox& B = m.box[b];
      /* robot bounding sphere against the box itself (not its bounding sphere: the boxes are flat slabs) */
      const V3 cb = v3((float)(bx - (double)B.pos[0]), (float)(by - (double)B.pos[1]), W.basez + zoff - B.pos[2]);
      V3 nn;
      near = sphere_box(matTvec(B.mat, cb), ld3(B.size), m.robot_radius, nn) < 0.05f;
    }
    cand[half] = ballot(near);
  }
}

/* Collision items against box b (wave-uniform): after the call lane `it` (position in con_order) holds the signed distance,
 * world normal and contact point (midway between the surfaces) of its item; false (and no barrier) when nothing is near.
 * (cg, rg): item_sphere of the lane's link geom.  zoff: extra height of the robot (lift loop). */
/* bounding sphere of link geom `lane`'s cloud in kernel coordinates (box independent: taken once per step / reset);
 * radius < 0: the lane has no geom, or not a calf geom when calf_only */
__device__ inline void item_sphere(const WaveMem& W, const GQ_MODEL GqDevModel& m, bool calf_only, V3& c, float& r) {
  const int lane = lane_id();
  c = v3(0.0f, 0.0f, 0.0f); r = -1.0f;
  if (lane < m.nlg) {
    const GQ_MODEL GqDevGeom& G = m.lg[lane];
    const bool calf = G.body > 0 && (G.body - 1) % 3 == 2;
    if (!calf_only || calf) {
      c = ld3(W.xpos[G.body]) + matvec(W.xmat[G.body], ld3(G.pos) + matvec(G.mat, ld3(G.aabb_c)));
      r = sqrtf(G.aabb_h[0] * G.aabb_h[0] + G.aabb_h[1] * G.aabb_h[1] + G.aabb_h[2] * G.aabb_h[2]) + G.radius;
    }
  }
}

__device__ inline bool box_item_scan(WaveMem& W, const GQ_MODEL GqDevModel& m, const GQ_MODEL float* vx, const GQ_MODEL float* vy, const GQ_MODEL float* vz, int b,
                                     double bx, double by, float zoff, V3 cg, float rg, float& dist, V3& nrm, V3& pt) {
  const int lane = lane_id();
  const GQ_MODEL GqDevBox& B = m.box[b];
  const V3 bp = v3((float)((double)B.pos[0] - bx), (float)((double)B.pos[1] - by), B.pos[2] - zoff); /* box relative to the base x/y */
  const V3 bs = ld3(B.size);
  const int nlg = m.nlg;
  /* phase A, lane = link geom: bounding spheres */
  bool needs = false;
  if (lane < nlg) {
    V3 nn; /* bounding sphere of the cloud against the box itself */
    needs = rg >= 0.0f && sphere_box(matTvec(B.mat, cg - bp), bs, rg, nn) < m.boxmix[B.cls][4 + lane].margin;
    if (!needs) W.u2.c.lg_dist[lane] = 1e30f;
  }
  uint64_t todo = ballot(needs);
  { /* nothing near this box (no link geom, no foot): skip the scan, the barrier and the item pass */
    bool foot_near = false;
    if (lane < 4) {
      V3 nn;
      foot_near = sphere_box(matTvec(B.mat, ld3(W.foot_world[lane]) - bp), bs, m.foot_radius[lane], nn) < m.boxmix[B.cls][lane].margin;
    }
    dist = 1e30f; nrm = v3(0.0f, 0.0f, 1.0f); pt = v3(0.0f, 0.0f, 0.0f);
    if ((todo | ballot(foot_near)) == 0) return false;
  }
  while (todo) { /* wave-uniform */
    const int g = ffs64(todo);
    todo &= todo - 1;
    const GQ_MODEL GqDevGeom& G = m.lg[g];
    const float* Rb = W.xmat[G.body];
    /* vertex -> box frame: p = A v + t, A = Bmat' Rb Rg, t = Bmat' (xpos + Rb gpos - bpos) */
    float RbRg[9], A[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) RbRg[3 * i + j] = Rb[3 * i] * G.mat[j] + Rb[3 * i + 1] * G.mat[3 + j] + Rb[3 * i + 2] * G.mat[6 + j];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) A[3 * i + j] = B.mat[i] * RbRg[j] + B.mat[3 + i] * RbRg[3 + j] + B.mat[6 + i] * RbRg[6 + j];
    const V3 og = ld3(W.xpos[G.body]) + matvec(Rb, ld3(G.pos)) - bp;
    const V3 t = matTvec(B.mat, og);
    float best = 1e30f;
    V3 bn = v3(0.0f, 0.0f, 1.0f), bc = v3(0.0f, 0.0f, 0.0f);
    for (int v0 = 0; v0 < G.cloud_num; v0 += GQ_WAVE) { /* wave-uniform trip count */
      const int i = G.cloud_adr + v0 + lane;
      const bool in = v0 + lane < G.cloud_num;
      const int ii = in ? i : G.cloud_adr;
      const V3 v = v3(vx[ii], vy[ii], vz[ii]);
      const V3 c = t + matvec(A, v);
      V3 n;
      const float dv = sphere_box(c, bs, G.radius, n);
      if (in && dv < best) { best = dv; bn = n; bc = c; }
    }
    const float wmin = wave_min(best);
    const int who = ffs64(ballot(best == wmin));
    const V3 n_l = v3(bcast(bn.x, who), bcast(bn.y, who), bcast(bn.z, who));
    const V3 c_l = v3(bcast(bc.x, who), bcast(bc.y, who), bcast(bc.z, who));
    if (lane == 0) {
      const V3 n_w = matvec(B.mat, n_l);
      const V3 v_w = bp + matvec(B.mat, c_l);
      W.u2.c.lg_dist[g] = wmin;
      st3(W.u2.c.lg_pt[g], v_w - (G.radius + 0.5f * wmin) * n_w);
      st3(GQ_BX_LGNRM(W) + 3 * g, n_w);
    }
  }
  wave_barrier();
  /* phase C, lane = collision item */
  dist = 1e30f; nrm = v3(0.0f, 0.0f, 1.0f); pt = v3(0.0f, 0.0f, 0.0f);
  if (lane < 4 + nlg) {
    const int code = m.con_order[lane];
    if (code < 4) {
      const V3 cw = ld3(W.foot_world[code]) - bp;
      V3 n_l;
      dist = sphere_box(matTvec(B.mat, cw), bs, m.foot_radius[code], n_l);
      nrm = matvec(B.mat, n_l);
      pt = ld3(W.foot_world[code]) - (m.foot_radius[code] + 0.5f * dist) * nrm;
    } else {
      dist = W.u2.c.lg_dist[code - 4]; nrm = ld3(GQ_BX_LGNRM(W) + 3 * (code - 4)); pt = ld3(W.u2.c.lg_pt[code - 4]);
    }
  }
  return true;
}

/* S6 (BOXES): append the contacts with the world boxes to the list the floor pass left in W (ncon, nefc, invalid,
 * foot_touch are updated; rows / row budget as in the floor pass).  Ends with a barrier. */
template <bool CONE>
__device__ inline void stage_box_contacts(WaveMem& W, const GQ_MODEL GqDevModel& m, const GQ_MODEL float* vx, const GQ_MODEL float* vy, const GQ_MODEL float* vz,
                                          double bx, double by, float mu_env) {
  const int lane = lane_id();
  const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  int ncon = uniform(W.ncon), rows = uniform(W.nefc), invalid = uniform(W.invalid), reserve = 0;
  int ft[4];
#pragma unroll
  for (int k = 0; k < 4; k++) ft[k] = uniform(W.foot_touch[k]);
  if (lane < ncon) { /* floor contacts: normal z, world geom = floor */
    st3(GQ_BX_CONNRM(W) + 3 * lane, v3(0.0f, 0.0f, 1.0f));
    GQ_BX_WCLS(W)[lane] = -1;
  }
  if constexpr (CONE)
    for (int c = 0; c < ncon; c++) { const int d = uniform(W.con_dim[c]); reserve += d > 1 ? d - 1 : 0; }
  uint64_t cand[2];
  box_candidates(W, m, bx, by, 0.0f, cand);
  V3 cg; float rg;
  item_sphere(W, m, false, cg, rg);
#pragma unroll 1
  for (int half = 0; half < 2; half++) {
    uint64_t todo = cand[half];
    while (todo) { /* wave-uniform */
      const int b = half * GQ_WAVE + ffs64(todo);
      todo &= todo - 1;
      float dist; V3 nrm, pt;
      if (!box_item_scan(W, m, vx, vy, vz, b, bx, by, 0.0f, cg, rg, dist, nrm, pt)) continue;
      const int cls = m.box[b].cls;
      bool touching = false, calf = false;
      int code = 0, body = 0, dim = 3;
      float mu = 0.0f;
      if (lane < 4 + m.nlg) {
        code = m.con_order[lane];
        const GQ_MODEL GqDevMix& X = m.boxmix[cls][code];
        touching = dist < X.margin;
        dim = X.dim;
        const float ff = m.boxcls_friction[cls][0]; /* _set_ground_friction leaves unnamed world boxes alone (quirk B8) */
        float fg;
        if (code < 4) { body = 3 + 3 * m.foot_leg[code]; calf = true; fg = mu_env >= 0.0f ? mu_env : m.foot_friction[code][0]; }
        else { const GQ_MODEL GqDevGeom& G = m.lg[code - 4]; body = G.body; calf = G.body > 0 && (G.body - 1) % 3 == 2; fg = G.friction[0]; }
        mu = fmaxf(1e-5f, X.rule == 0 ? fmaxf(ff, fg) : (X.rule == 1 ? ff : fg));
      }
      const uint64_t touch_mask = ballot(touching);
      if (touch_mask == 0) { wave_barrier(); continue; }
      invalid |= ballot(touching && !calf) != 0;
#pragma unroll
      for (int k = 0; k < 4; k++) ft[k] |= ballot(touching && body == 3 + 3 * m.foot_leg[k]) != 0;
      const int idx = ncon + popc64(touch_mask & lt);
      const bool kept = touching && idx < GQ_MAXCON;
      const int need = dim == 1 ? 1 : (CONE ? dim : 2 * (dim - 1));
      const uint64_t m1 = ballot(kept && need == 1), m3 = ballot(kept && need == 3), m4 = ballot(kept && need == 4), m6 = ballot(kept && need == 6);
      const int row0 = rows + popc64(m1 & lt) + 3 * popc64(m3 & lt) + 4 * popc64(m4 & lt) + 6 * popc64(m6 & lt);
      const int res = CONE ? reserve + 2 * popc64(m3 & lt) + 5 * popc64(m6 & lt) + (need > 1 ? need - 1 : 0) : 0;
      const bool fits = kept && row0 + need + res <= (CONE ? 64 : GQ_MAXEFC) && row0 + need <= GQ_MAXEFC;
      const uint64_t f1 = ballot(fits && need == 1), f3 = ballot(fits && need == 3), f4 = ballot(fits && need == 4), f6 = ballot(fits && need == 6);
      if (fits) {
        const GQ_MODEL GqDevMix& X = m.boxmix[cls][code];
        W.con_geom[idx] = code; W.con_body[idx] = body; W.con_dim[idx] = dim; W.con_row[idx] = row0;
        W.con_dist[idx] = dist; W.con_inc[idx] = X.includemargin; W.con_mu[idx] = mu;
        st3(W.con_pos[idx], pt);
        W.con_solref[idx][0] = X.solref[0]; W.con_solref[idx][1] = X.solref[1];
#pragma unroll
        for (int q = 0; q < 5; q++) W.con_solimp[idx][q] = X.solimp[q];
        st3(GQ_BX_CONNRM(W) + 3 * idx, nrm);
        GQ_BX_WCLS(W)[idx] = cls;
      }
      ncon += popc64(f1 | f3 | f4 | f6);
      rows += popc64(f1) + 3 * popc64(f3) + 4 * popc64(f4) + 6 * popc64(f6);
      if constexpr (CONE) reserve += 2 * popc64(f3) + 5 * popc64(f6);
      wave_barrier();
    }
  }
  if (lane == 0) {
    W.ncon = ncon; W.nefc = rows; W.invalid = invalid;
#pragma unroll
    for (int k = 0; k < 4; k++) W.foot_touch[k] = ft[k];
  }
  wave_barrier();
}

}  // namespace gq
