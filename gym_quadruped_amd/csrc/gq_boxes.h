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
#include "gq_pairs.h"
#include "gq_convex.h"
#include "gq_exchange.h"

namespace gq {

/* LDS scratch carved out of WaveMem::F (Newton path: F[0][0..17] = h*damping, the rest is free) */
#define GQ_BX_WCLS(W) (reinterpret_cast<int32_t*>(&(W).F[0][32]))   /* [12] world geom of contact c: -1 floor, else box class */
#define GQ_BX_LGNRM(W) (&(W).F[1][0])                                  /* [GQ_MAXLG][3] normal of the geom's hit on the current box */
#define GQ_BX_CONNRM(W) (&(W).F[1][3 * GQ_MAXLG])                      /* [12][3] contact normals */
/* bounding spheres of the link geoms' clouds (item_sphere), [GQ_MAXLG][4]: parked in the (idle until S7) J block behind the region the
 * kinematics scratch and the self-collision tables use, and re-read per world box - four registers less across the box loop */
#define GQ_BX_ISPH(W) (&(W).u.B[40][0])
/* oriented bounding boxes of the link geoms' clouds in kernel coordinates, [GQ_MAXLG][GQ_BX_OBBW]: centre, the three half-axis vectors, the geom's
 * radius (stage_box_contacts writes them once per step; rows 0 - 27 of the J block, free until the self-collision pass behind the box loop) */
#define GQ_BX_OBBW 13
#define GQ_BX_OBB(W) (&(W).u.B[0][0])
/* scratch of the convex routine (gq_convex.h): the two shape descriptors + the result in the factor block behind GQ_BX_WCLS (48 words, both
 * phases); its polytope (200 words) in rows 28 - 39 of the J block during the world-box loop (between the clouds' boxes and GQ_BX_ISPH) and in
 * rows 38 - 49 during the self-collision pass (behind its end-point table and pair list; GQ_BX_ISPH / GQ_BX_PSPH are dead by then) */
#define GQ_CVX_SHP(W) ((LdsF)&(W).F[0][44])
#define GQ_CVX_POLY_BOX(W) ((LdsF)&(W).u.B[28][0])
#define GQ_CVX_POLY_SELF(W) ((LdsF)&(W).u.B[38][0])
static_assert(GQ_BX_OBBW * GQ_MAXLG <= 28 * GQ_NVD && 28 * GQ_NVD + GQ_CVX_POLY_WORDS <= 40 * GQ_NVD && 38 * GQ_NVD + GQ_CVX_POLY_WORDS <= 64 * GQ_NVD && 44 + GQ_CVX_SHP_WORDS <= GQ_FACTOR_SIZE,
              "scratch of the convex routine");

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

/* bit mask (two words) of the boxes that may touch a collision item: lane = box; first the robot's bounding sphere against
 * the box (cheap, most boxes of a scene fail it), then - for the lanes that passed - the bounding sphere of every item (feet,
 * link geoms; (cg, rg) = item_sphere of lane = geom, broadcast in a wave-uniform loop) against the box.  The per-box work
 * of the callers is a serial, latency-bound loop (model loads, scans, barriers): a dense box scene puts 10-18 boxes inside
 * the robot's sphere but only 2-5 near an item.  base = (0, 0, basez) in kernel coordinates; zoff lifts the robot. */
__device__ inline void box_candidates(const WaveMem& W, const GQ_MODEL GqDevModel& m, double bx, double by, float zoff, uint64_t cand[2], V3 cg, float rg) {
  const int lane = lane_id();
  const int nlg = m.nlg;
#pragma unroll 1
  for (int half = 0; half < 2; half++) {
    const int b = half * GQ_WAVE + lane;
    bool near = false;
    V3 bp = v3(0.0f, 0.0f, 0.0f), bs = v3(1.0f, 1.0f, 1.0f);
    float Bm[9] = {1.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 0.0f, 1.0f};
    if (b < m.nbox) {
      const GQ_MODEL GqDevBox& B = m.box[b];
      /* robot bounding sphere against the box itself (not its bounding sphere: the boxes are flat slabs) */
      bp = v3((float)((double)B.pos[0] - bx), (float)((double)B.pos[1] - by), B.pos[2] - zoff);
      bs = ld3(B.size);
#pragma unroll
      for (int i = 0; i < 9; i++) Bm[i] = B.mat[i];
      V3 nn;
      near = sphere_box(matTvec(Bm, v3(0.0f, 0.0f, W.basez) - bp), bs, m.robot_radius, nn) < 0.05f;
    }
    uint64_t coarse = ballot(near);
    if (coarse != 0) { /* wave-uniform */
      bool hit = false;
      for (int k = 0; k < 4; k++) { /* feet */
        V3 nn;
        hit = hit || sphere_box(matTvec(Bm, ld3(W.foot_world[k]) - bp), bs, m.foot_radius[k], nn) < 0.05f;
      }
      for (int k = 0; k < nlg; k++) { /* link geoms */
        const float rk = bcast(rg, k);
        if (rk < 0.0f) continue; /* wave-uniform: no geom / not a calf geom in the lift loop */
        const V3 ck = v3(bcast(cg.x, k), bcast(cg.y, k), bcast(cg.z, k));
        V3 nn;
        hit = hit || sphere_box(matTvec(Bm, ck - bp), bs, rk, nn) < 0.05f;
      }
      coarse = ballot(near && hit);
    }
    cand[half] = coarse;
  }
}

/* oriented bounding box of link geom `lane`'s cloud (geom-frame AABB) in kernel coordinates -> GQ_BX_OBB (box independent: once per step / reset;
 * the caller's barrier comes before the first box_item_scan<.., OBB = true>) */
__device__ inline void item_obb_store(WaveMem& W, const GQ_MODEL GqDevModel& m) {
  const int lane = lane_id();
  if (lane < m.nlg) {
    const GQ_MODEL GqDevGeom& G = m.lg[lane];
    const float* Rb = W.xmat[G.body];
    float RbRg[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) RbRg[3 * i + j] = Rb[3 * i] * G.mat[j] + Rb[3 * i + 1] * G.mat[3 + j] + Rb[3 * i + 2] * G.mat[6 + j];
    float* O = GQ_BX_OBB(W) + GQ_BX_OBBW * lane;
    st3(O, ld3(W.xpos[G.body]) + matvec(Rb, ld3(G.pos)) + matvec(RbRg, ld3(G.aabb_c)));
#pragma unroll
    for (int j = 0; j < 3; j++) st3(O + 3 + 3 * j, G.aabb_h[j] * v3(RbRg[j], RbRg[3 + j], RbRg[6 + j]));
    O[12] = G.radius;
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

/* PL: what is kept of the item record of lane `it` (prim_lane); H: the item's contact candidates with box b - one for a foot
 * sphere or a hull / cylinder cloud (its deepest inflated vertex), up to 2 / 4 for the robot's sphere / capsule / box geoms
 * (exact pair routines, gq_pairs.h).  PRIM false: the model has no such geom and the routines are not compiled in. */
template <bool PRIM, int BATCH = 2 /* chunks whose vertex loads go out together (below) */, bool OBB = false /* GQ_BX_OBB holds the geoms' boxes */>
__device__ inline bool box_item_scan(WaveMem& W, const GQ_MODEL GqDevModel& m, const GQ_MODEL float* vx, const GQ_MODEL float* vy, const GQ_MODEL float* vz, int b,
                                     double bx, double by, float zoff, V3 cg, float rg, const PrimLane& PL, PairHit& H) {
  float dist; V3 nrm, pt;
  const int lane = lane_id();
  const GQ_MODEL GqDevBox& B = m.box[b];
  const V3 bp = v3((float)((double)B.pos[0] - bx), (float)((double)B.pos[1] - by), B.pos[2] - zoff); /* box relative to the base x/y */
  const V3 bs = ld3(B.size);
  const int nlg = m.nlg;
  /* phase A, lane = link geom: bounding spheres */
  bool needs = false;
  V3 hintw = v3(0.0f, 0.0f, 0.0f); /* lane = geom: out of the box towards the cloud's centre - GJK's first direction (a leg above a wide box: the box's top normal, which
                                    * separates the two at the first support query; the line of centres would be nearly horizontal) */
  if (lane < nlg) {
    V3 nn; /* bounding sphere of the cloud against the box itself */
    const float marg = m.boxmix[B.cls][4 + lane].margin;
    needs = rg >= 0.0f && PL.cloud && sphere_box(matTvec(B.mat, cg - bp), bs, rg, nn) < marg;
    hintw = matvec(B.mat, nn);
    if constexpr (OBB) if (needs) {
      /* the sphere of a long thin link is loose: half of the (geom, box) pairs it lets through have no vertex near the box, and each costs the
       * transform, the chunk-box fetch and its round trip before that is known.  The cloud's box (geom-frame AABB, taken to kernel coordinates
       * once per step) against the world box, separating along the world box's axes: gap_i = |c_i| - s_i - sum_j |e_j . b_i| */
      const float* O = GQ_BX_OBB(W) + GQ_BX_OBBW * lane;
      const V3 c = matTvec(B.mat, ld3(O) - bp);
      const V3 e0 = matTvec(B.mat, ld3(O + 3)), e1 = matTvec(B.mat, ld3(O + 6)), e2 = matTvec(B.mat, ld3(O + 9));
      const V3 ee = v3(fabsf(e0.x) + fabsf(e1.x) + fabsf(e2.x), fabsf(e0.y) + fabsf(e1.y) + fabsf(e2.y), fabsf(e0.z) + fabsf(e1.z) + fabsf(e2.z));
      const V3 gap = v3(fmaxf(0.0f, fabsf(c.x) - bs.x - ee.x), fmaxf(0.0f, fabsf(c.y) - bs.y - ee.y), fmaxf(0.0f, fabsf(c.z) - bs.z - ee.z));
      needs = sqrtf(dot(gap, gap)) - O[12] < marg + 1e-5f;
    }
    if (needs) { /* third test, still ONE lane: the hull itself (its support-function grid, cvx_hgrid) against the box along the direction GJK would
                  * try first - nine in ten (geom, box) pairs that pass the boxes' test end at that query of the wave-serial routine (hyqreal1 over a box
                  * field: 31 of them per env-step, tools/convex_census.py) */
      const GQ_MODEL GqDevGeom& G = m.lg[lane];
      if (G.hgrid_adr >= 0 && dot(hintw, hintw) > 0.25f) {
        const float* Rb = W.xmat[G.body];
        float Rg[9];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
          for (int j = 0; j < 3; j++) Rg[3 * i + j] = Rb[3 * i] * G.mat[j] + Rb[3 * i + 1] * G.mat[3 + j] + Rb[3 * i + 2] * G.mat[6 + j];
        const V3 tg = ld3(W.xpos[G.body]) + matvec(Rb, ld3(G.pos));
        const V3 d = hintw, dl = matTvec(Rg, -1.0f * d), db = matTvec(B.mat, d);
        const float hB = -dot(tg, d) + cvx_hgrid(vx, G.hgrid_adr, dl);                                    /* max over the hull of v . (-d) */
        const float hA = dot(bp, d) + bs.x * fabsf(db.x) + bs.y * fabsf(db.y) + bs.z * fabsf(db.z);          /* max over the box of v . d */
        if (-hA - hB > (marg + G.radius + 1e-5f) * fast_sqrt(dot(d, d))) needs = false;
      }
    }
    if (!needs) W.u2.c.lg_dist[lane] = 1e30f;
  }
  uint64_t todo = ballot(needs);
  /* primitive link geoms (lane = item): bounding sphere of the item against the box; the exact routine runs in phase C */
  bool prim_near = false;
  if constexpr (PRIM) {
    if (PL.ptype > 0) {
      V3 nn;
      prim_near = sphere_box(matTvec(B.mat, ld3(PL.sph) - bp), bs, PL.sph[3], nn) < PL.margin + m.boxmix[B.cls][PL.code].margin;
    }
  }
  H.n = 0;
  { /* nothing near this box (no link geom, no foot): skip the scan, the barrier and the item pass */
    bool foot_near = false;
    if (lane < 4) {
      V3 nn;
      foot_near = sphere_box(matTvec(B.mat, ld3(W.foot_world[lane]) - bp), bs, m.foot_radius[lane], nn) < m.boxmix[B.cls][lane].margin;
    }
    dist = 1e30f; nrm = v3(0.0f, 0.0f, 1.0f); pt = v3(0.0f, 0.0f, 0.0f);
    if ((todo | ballot(foot_near) | ballot(prim_near)) == 0) return false;
  }
  if (__builtin_expect(todo != 0, 0)) { /* wave-uniform: the world box as shape A of the convex routine (mjc_Convex: box = geom 1, the normal points out of it) */
    CvxShape A;
    A.kind = 1; A.adr = 0; A.num = 0; A.pm = -1; A.t = bp; A.h = bs; A.r = 0.0f;
#pragma unroll
    for (int i = 0; i < 9; i++) A.R[i] = B.mat[i];
    wave_barrier();
    cvx_shape_store(GQ_CVX_SHP(W), A);
  }
  while (__builtin_expect(todo != 0, 0)) { /* wave-uniform: one hull / cylinder cloud against the box - GJK + EPA on the wavefront (gq_convex.h) */
    const int g = ffs64(todo);
    todo &= todo - 1;
    const GQ_MODEL GqDevGeom& G = m.lg[g];
    const float* Rb = W.xmat[G.body];
    CvxShape S;
    S.kind = 0; S.adr = G.plane_adr; S.num = G.cloud_num; S.pm = G.cap_adr; S.r = G.radius; S.h = v3(0.0f, 0.0f, 0.0f);
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) S.R[3 * i + j] = Rb[3 * i] * G.mat[j] + Rb[3 * i + 1] * G.mat[3 + j] + Rb[3 * i + 2] * G.mat[6 + j];
    S.t = ld3(W.xpos[G.body]) + matvec(Rb, ld3(G.pos));
    wave_barrier();
    cvx_shape_store(GQ_CVX_SHP(W) + GQ_CVX_SHAPE_WORDS, S);
    wave_barrier();
    const bool hit = cvx_pair_wave(GQ_CVX_SHP(W), GQ_CVX_POLY_BOX(W), vx, vy, vz, m.boxmix[B.cls][4 + g].margin, v3(bcast(hintw.x, g), bcast(hintw.y, g), bcast(hintw.z, g)));
    if (lane == 0) {
      LdsCF out = GQ_CVX_SHP(W) + 2 * GQ_CVX_SHAPE_WORDS;
      W.u2.c.lg_dist[g] = hit ? out[0] : 1e30f;
      st3(GQ_BX_LGNRM(W) + 3 * g, ld3(out + 1));
      st3(W.u2.c.lg_pt[g], ld3(out + 4));
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
    } else if (!PRIM || PL.ptype == 0) {
      dist = W.u2.c.lg_dist[code - 4]; nrm = ld3(GQ_BX_LGNRM(W) + 3 * (code - 4)); pt = ld3(W.u2.c.lg_pt[code - 4]);
    }
  }
  H.n = dist < 1e29f ? 1 : 0; H.dist[0] = dist; H.nrm[0] = nrm; H.pos[0] = pt;
  if constexpr (PRIM) {
    if (prim_near) { /* the robot's sphere / capsule / box geoms against the box: exact (mjc_SphereBox / CapsuleBox / BoxBox geometry) */
      const GQ_MODEL GqDevItem& I = m.item[lane];
      const float marg = m.boxmix[B.cls][PL.code].margin;
      float Rw[9];
#pragma unroll
      for (int i = 0; i < 9; i++) Rw[i] = B.mat[i];
      const float* Rb = W.xmat[I.body];
      float A[9];
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) A[3 * i + j] = Rb[3 * i] * I.mat[j] + Rb[3 * i + 1] * I.mat[3 + j] + Rb[3 * i + 2] * I.mat[6 + j];
      const V3 pc = ld3(PL.sph);
      if (PL.ptype == 6) box_box(bp, Rw, bs, pc, A, v3(I.psize[0], I.psize[1], I.psize[2]), marg, H);
      else {
        const V3 ax = PL.ptype == 3 ? I.psize[1] * v3(A[2], A[5], A[8]) : v3(0.0f, 0.0f, 0.0f);
        capsule_box(pc - ax, pc + ax, I.psize[0], bp, Rw, bs, marg, H);
      }
    }
  }
  return true;
}


/* ------------------------------------------------------------------ height field (terrain.py add_perlin_heightfield :26-113)
 * One static hfield geom: nrow x ncol elevations over [-sx, sx] x [-sy, sy] around hf_pos, identity orientation.  Every
 * grid cell is two triangles, split like MuJoCo's prism strip (vertices (c,r), (c,r+1), (c+1,r), (c+1,r+1): the diagonal
 * runs from (c,r+1) to (c+1,r)).  MuJoCo collides a convex geom with the prisms under its bounding box, one contact per
 * prism; here every collision item keeps ONE contact with the height field: a foot sphere its closest triangle (exact
 * point-triangle distance, centre above the surface), a link geom its deepest cloud vertex measured against the plane
 * of the triangle under that vertex - the same simplification as for the world boxes, restated in the oracle. */
struct HfTri { V3 a, b, c, n; }; /* corners and unit normal (n.z > 0) */
/* All height-field geometry is expressed relative to a reference grid corner (cb, rb) next to the robot's base, so that
 * the fp32 coordinates stay small (a foot sphere of 2-3 cm resolved against coordinates of 15 m would lose its normal to
 * round-off).  Cell (cb + kc, rb + kr) has its (c, r) corner at (kc dx, kr dy). */
struct HfRef { int cb, rb; };

/* triangle `upper` (0: contains the cell's (c,r) corner, 1: contains (c+1,r+1)) of cell (ref + (kc, kr)); false outside the grid */
__device__ __forceinline__ bool hf_cell_triangle(const GQ_MODEL GqDevModel& m, const GQ_MODEL float* H, HfRef ref, int kc, int kr, int upper, HfTri& t) {
  const int nc = m.hf_ncol, c = ref.cb + kc, r = ref.rb + kr;
  if (c < 0 || r < 0 || c > nc - 2 || r > m.hf_nrow - 2) return false;
  const float x0 = m.hf_dx * (float)kc, y0 = m.hf_dy * (float)kr, x1 = x0 + m.hf_dx, y1 = y0 + m.hf_dy;
  const float h10 = H[r * nc + c + 1], h01 = H[(r + 1) * nc + c], hq = upper ? H[(r + 1) * nc + c + 1] : H[r * nc + c];
  t.b = v3(x1, y0, h10); t.c = v3(x0, y1, h01);
  float gx, gy;
  if (!upper) { t.a = v3(x0, y0, hq); gx = (h10 - hq) * m.hf_inv_dx; gy = (h01 - hq) * m.hf_inv_dy; }
  else { t.a = v3(x1, y1, hq); gx = (hq - h01) * m.hf_inv_dx; gy = (hq - h10) * m.hf_inv_dy; }
  const float inv = fast_rsqrt(gx * gx + gy * gy + 1.0f);
  t.n = v3(-gx * inv, -gy * inv, inv);
  return true;
}
/* the triangle under the point (x, y) given relative to the reference corner; false outside the grid */
__device__ __forceinline__ bool hf_triangle_under(const GQ_MODEL GqDevModel& m, const GQ_MODEL float* H, HfRef ref, float x, float y, HfTri& t) {
  const float fx = x * m.hf_inv_dx, fy = y * m.hf_inv_dy;
  const int kc = (int)floorf(fx), kr = (int)floorf(fy);
  return hf_cell_triangle(m, H, ref, kc, kr, (fx - (float)kc) + (fy - (float)kr) > 1.0f ? 1 : 0, t);
}
/* closest point of triangle (a, b, c) to p (Ericson, Real-Time Collision Detection 5.1.5) */
__device__ inline V3 closest_on_triangle(V3 p, V3 a, V3 b, V3 c, bool& interior) {
  interior = false;
  const V3 ab = b - a, ac = c - a, ap = p - a;
  const float d1 = dot(ab, ap), d2 = dot(ac, ap);
  if (d1 <= 0.0f && d2 <= 0.0f) return a;
  const V3 bp = p - b;
  const float d3 = dot(ab, bp), d4 = dot(ac, bp);
  if (d3 >= 0.0f && d4 <= d3) return b;
  const float vc = d1 * d4 - d3 * d2;
  if (vc <= 0.0f && d1 >= 0.0f && d3 <= 0.0f) return a + (d1 / (d1 - d3)) * ab;
  const V3 cp = p - c;
  const float d5 = dot(ab, cp), d6 = dot(ac, cp);
  if (d6 >= 0.0f && d5 <= d6) return c;
  const float vb = d5 * d2 - d1 * d6;
  if (vb <= 0.0f && d2 >= 0.0f && d6 <= 0.0f) return a + (d2 / (d2 - d6)) * ac;
  const float va = d3 * d6 - d5 * d4;
  if (va <= 0.0f && (d4 - d3) >= 0.0f && (d5 - d6) >= 0.0f) return b + ((d4 - d3) / ((d4 - d3) + (d5 - d6))) * (c - b);
  const float den = 1.0f / (va + vb + vc);
  interior = true; /* the projection of p falls inside the triangle */
  return a + (vb * den) * ab + (vc * den) * ac;
}
/* sphere (centre p relative to the reference corner, radius r) against the height field: signed distance, normal,
 * false if nothing within `reach` of the surface.  Triangles of the cells under the sphere's footprint (+ reach) are visited. */
/* one triangle of the scan below: the sphere's signed distance to it and the normal, false if the triangle has no say */
__device__ __forceinline__ bool sphere_hf_triangle(const GQ_MODEL GqDevModel& m, const GQ_MODEL float* H, HfRef ref, V3 p, float r, int cc, int rr, int up, float& dd, V3& nn) {
  HfTri t;
  if (!hf_cell_triangle(m, H, ref, cc, rr, up, t)) return false;
  const float side = dot(p - t.a, t.n);
  bool inside;
  const V3 q = closest_on_triangle(p, t.a, t.b, t.c, inside);
  if (inside) { dd = side - r; nn = t.n; return true; } /* over (or under) the face: distance along its normal, whatever the sign */
  if (side < 0.0f) return false;                        /* below the plane and outside the column: a neighbour's business */
  const V3 d = p - q;
  const float l2 = dot(d, d);
  if (!(l2 > 1e-12f)) return false;
  const float inv = fast_rsqrt(l2);
  dd = l2 * inv - r; nn = inv * d;
  return true;
}
__device__ inline bool sphere_hfield(const GQ_MODEL GqDevModel& m, const GQ_MODEL float* H, HfRef ref, V3 p, float r, float reach, float& dist, V3& n) {
  const float R = r + reach;
  const int c0 = (int)floorf((p.x - R) * m.hf_inv_dx), c1 = (int)floorf((p.x + R) * m.hf_inv_dx);
  const int r0 = (int)floorf((p.y - R) * m.hf_inv_dy), r1 = (int)floorf((p.y + R) * m.hf_inv_dy);
  dist = 1e30f; n = v3(0.0f, 0.0f, 1.0f);
  for (int rr = r0; rr <= r1; rr++)
    for (int cc = c0; cc <= c1; cc++)
      for (int up = 0; up < 2; up++) {
        float dd; V3 nn;
        if (sphere_hf_triangle(m, H, ref, p, r, cc, rr, up, dd, nn) && dd < dist) { dist = dd; n = nn; }
      }
  return dist < reach;
}

/* Collision items against the height field; same contract as box_item_scan (lane `it` of con_order gets its item's
 * distance / normal / point, false and no barrier when nothing is near). */
__device__ inline bool hfield_item_scan(WaveMem& W, const GQ_MODEL GqDevModel& m, const GQ_MODEL float* vx, const GQ_MODEL float* vy, const GQ_MODEL float* vz,
                                        double bx, double by, float zoff, V3 cg, float rg, float& dist, V3& nrm, V3& pt) {
  const int lane = lane_id();
  const GQ_MODEL float* H = mptr(m.hf_data);
  /* reference corner: the grid node at or below the base x/y (f64), its position relative to the base in kernel coordinates */
  const double gx = (bx - (double)m.hf_pos[0] + (double)m.hf_sx) * (double)m.hf_inv_dx, gy = (by - (double)m.hf_pos[1] + (double)m.hf_sy) * (double)m.hf_inv_dy;
  const HfRef ref = {(int)floor(gx), (int)floor(gy)};
  const V3 hp = v3((float)((double)m.hf_pos[0] - (double)m.hf_sx + (double)ref.cb * (double)m.hf_dx - bx),
                   (float)((double)m.hf_pos[1] - (double)m.hf_sy + (double)ref.rb * (double)m.hf_dy - by), m.hf_pos[2] - zoff);
  const int nlg = m.nlg, cls = m.hf_cls;
  /* phase A, lane = link geom / foot: bounding sphere above the surface?  (surface within rho of a point rises at most maxslope * rho) */
  /* (lanes 60-63 take the feet - GQ_MAXLG < 60 -: ONE pass through the elevation loads for geoms and feet - they were two, a memory round trip each) */
  bool needs = false;
  {
    const bool isfoot = lane >= 60, isgeom = lane < nlg && rg >= 0.0f;
    const int fk = isfoot ? lane - 60 : 0;
    const V3 cl = (isfoot ? ld3(W.foot_world[fk]) : cg) - hp;
    const float rb = isfoot ? m.foot_radius[fk] : rg;
    HfTri t;
    if ((isfoot || isgeom) && hf_triangle_under(m, H, ref, cl.x, cl.y, t)) {
      const float hc = t.a.z - (t.n.x * (cl.x - t.a.x) + t.n.y * (cl.y - t.a.y)) / t.n.z;
      needs = cl.z - rb - (hc + m.hf_maxslope * rb) < m.boxmix[cls][isfoot ? fk : 4 + lane].margin;
    }
  }
  if (lane < nlg && !needs) W.u2.c.lg_dist[lane] = 1e30f;
  const uint64_t near_all = ballot(needs);
  uint64_t todo = near_all & ((1ull << 60) - 1ull);
  const uint64_t feet = near_all >> 60;
  dist = 1e30f; nrm = v3(0.0f, 0.0f, 1.0f); pt = v3(0.0f, 0.0f, 0.0f);
  if ((todo | feet) == 0) return false;
  if (m.self_cut == 10) return false; /* profiling aid (GQ_SELF_CUT): 10 stop after the bounding tests, 11 no flattened pass, 12 no serial scans, 13 no foot narrow phase */
  if (m.self_cut == 11) todo &= ~m.flat_mask;
  if (m.self_cut == 12) todo &= m.flat_mask;
  /* small clouds (boxes, capsules: a handful of vertices each) are evaluated FLATTENED: lane = (geom, vertex) slot of the host
   * table, every needed geom of a pass at once - one memory latency per pass for vertex, cell and elevations instead of one
   * per geom, and no 64-lane scan of 8 vertices; then lane = geom picks the deepest of its slots (first in vertex order, as
   * the serial scan does) out of LDS and fetches that slot's normal and point with ds_bpermute */
  if (todo & m.flat_mask) { /* wave-uniform */
    float* S = W.force; /* 64 floats of scratch: the solver's row forces do not exist yet */
    const uint64_t flat = todo & m.flat_mask;
    todo &= ~m.flat_mask;
    for (int p0 = 0; p0 < m.flat_n; p0 += GQ_WAVE) { /* wave-uniform */
      const int slot = p0 + lane;
      const int g = slot < m.flat_n ? (int)m.flat_geom[slot] : 255;
      const bool live = g != 255 && ((flat >> (g & 63)) & 1ull);
      float dv = 1e30f;
      V3 bn = v3(0.0f, 0.0f, 1.0f), bc = v3(0.0f, 0.0f, 0.0f);
      if (live) {
        const GQ_MODEL GqDevGeom& G = m.lg[g];
        const int iv = m.flat_vert[slot];
        const float* Rb = W.xmat[G.body];
        const V3 u = ld3(G.pos) + matvec(G.mat, v3(vx[iv], vy[iv], vz[iv]));
        const V3 c = ld3(W.xpos[G.body]) + matvec(Rb, u) - hp;
        HfTri t;
        if (hf_triangle_under(m, H, ref, c.x, c.y, t)) { dv = dot(c - t.a, t.n) - G.radius; bn = t.n; bc = c; }
      }
      S[lane] = dv;
      wave_barrier();
      float best = 1e30f;
      int arg = lane;
      bool mine = false;
      if (lane < nlg) {
        const GQ_MODEL GqDevGeom& G = m.lg[lane];
        mine = ((flat >> lane) & 1ull) && G.flat_adr >= p0 && G.flat_adr < p0 + GQ_WAVE;
        if (mine) {
          const int a0 = G.flat_adr - p0;
          for (int k = 0; k < G.cloud_num; k++) {
            const float dk = S[a0 + k];
            if (dk < best) { best = dk; arg = a0 + k; }
          }
        }
      }
      const V3 n_w = v3(shfl_idx(bn.x, arg), shfl_idx(bn.y, arg), shfl_idx(bn.z, arg));
      const V3 c_w = v3(shfl_idx(bc.x, arg), shfl_idx(bc.y, arg), shfl_idx(bc.z, arg));
      if (mine) {
        const GQ_MODEL GqDevGeom& G = m.lg[lane];
        W.u2.c.lg_dist[lane] = best;
        st3(W.u2.c.lg_pt[lane], hp + c_w - (G.radius + 0.5f * best) * n_w);
        st3(GQ_BX_LGNRM(W) + 3 * lane, n_w);
      }
      wave_barrier(); /* S is rewritten by the next pass */
    }
  }
  while (todo) { /* wave-uniform */
    const int g = ffs64(todo);
    todo &= todo - 1;
    const GQ_MODEL GqDevGeom& G = m.lg[g];
    const float* Rb = W.xmat[G.body];
    float A[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) A[3 * i + j] = Rb[3 * i] * G.mat[j] + Rb[3 * i + 1] * G.mat[3 + j] + Rb[3 * i + 2] * G.mat[6 + j];
    const V3 t0 = ld3(W.xpos[G.body]) + matvec(Rb, ld3(G.pos)) - hp;
    float best = 1e30f;
    V3 bn = v3(0.0f, 0.0f, 1.0f), bc = v3(0.0f, 0.0f, 0.0f);
    for (int v0 = 0; v0 < G.cloud_num; v0 += GQ_WAVE) { /* wave-uniform trip count */
      const int i = G.cloud_adr + v0 + lane;
      const bool in = v0 + lane < G.cloud_num;
      const int ii = in ? i : G.cloud_adr;
      const V3 c = t0 + matvec(A, v3(vx[ii], vy[ii], vz[ii]));
      HfTri t;
      if (in && hf_triangle_under(m, H, ref, c.x, c.y, t)) {
        const float dv = dot(c - t.a, t.n) - G.radius;
        if (dv < best) { best = dv; bn = t.n; bc = c; }
      }
    }
    const float wmin = wave_min(best);
    const int who = ffs64(ballot(best == wmin));
    const V3 n_w = v3(bcast(bn.x, who), bcast(bn.y, who), bcast(bn.z, who));
    const V3 c_w = v3(bcast(bc.x, who), bcast(bc.y, who), bcast(bc.z, who));
    if (lane == 0) {
      W.u2.c.lg_dist[g] = wmin;
      st3(W.u2.c.lg_pt[g], hp + c_w - (G.radius + 0.5f * wmin) * n_w);
      st3(GQ_BX_LGNRM(W) + 3 * g, n_w);
    }
  }
  wave_barrier();
  /* the feet: a sphere against the triangles of the cells under its footprint (sphere_hfield).  A footprint of a few centimetres on cells of
   * tens of centimetres covers one cell, two or four when it straddles a grid line: lane = (foot, cell of the 2 x 2 block, triangle) evaluates
   * the <= 8 triangles of every near foot at once - one pass through the elevation loads instead of up to eight dependent ones on the foot's lane,
   * which all four feet walked in lockstep (the longest footprint set the trip count) - and the foot takes the first of the nearest in the
   * serial scan's order (rows, columns, lower / upper triangle).  A footprint over more than 2 x 2 cells (a fine grid): the serial scan. */
  bool feet_par = false;
  float fdd = 1e30f;
  V3 fnn = v3(0.0f, 0.0f, 1.0f);
  if (feet != 0 && m.self_cut != 13) { /* wave-uniform */
    const int fk = (lane >> 3) & 3, k = lane & 7;
    const bool live = lane < 32 && ((feet >> fk) & 1ull);
    const V3 p = ld3(W.foot_world[fk]) - hp;
    const float r = m.foot_radius[fk], R = r + fmaxf(m.boxmix[cls][fk].margin, 0.0f) + 1e-4f;
    const int c0 = (int)floorf((p.x - R) * m.hf_inv_dx), c1 = (int)floorf((p.x + R) * m.hf_inv_dx);
    const int r0 = (int)floorf((p.y - R) * m.hf_inv_dy), r1 = (int)floorf((p.y + R) * m.hf_inv_dy);
    feet_par = ballot(live && (c1 - c0 > 1 || r1 - r0 > 1)) == 0;
    if (feet_par) {
      const int cc = c0 + ((k >> 1) & 1), rr = r0 + ((k >> 2) & 1);
      float dd; V3 nn;
      if (live && cc <= c1 && rr <= r1 && sphere_hf_triangle(m, H, ref, p, r, cc, rr, k & 1, dd, nn)) { fdd = dd; fnn = nn; }
      float* S = W.force; /* scratch, as in the flattened pass */
      S[lane] = fdd;
      wave_barrier();
    }
  }
  /* phase C, lane = collision item */
  int src = lane;
  float fbest = 1e30f;
  const int code = lane < 4 + nlg ? (int)m.con_order[lane] : 4;
  const bool myfoot = code < 4 && ((feet >> code) & 1ull) && m.self_cut != 13;
  if (feet_par && myfoot) {
    const float* S = W.force + 8 * code;
#pragma unroll
    for (int k = 0; k < 8; k++) { const float dk = S[k]; if (dk < fbest) { fbest = dk; src = 8 * code + k; } }
  }
  if (feet_par) { /* wave-uniform: the winner's normal */
    const V3 n = v3(shfl_idx(fnn.x, src), shfl_idx(fnn.y, src), shfl_idx(fnn.z, src));
    if (myfoot && fbest < fmaxf(m.boxmix[cls][code].margin, 0.0f) + 1e-4f) {
      dist = fbest; nrm = n; pt = ld3(W.foot_world[code]) - (m.foot_radius[code] + 0.5f * fbest) * n;
    }
    wave_barrier(); /* W.force is scratch no longer */
  }
  if (lane < 4 + nlg) {
    if (code < 4) {
      if (myfoot && !feet_par) {
        V3 n; float d;
        if (sphere_hfield(m, H, ref, ld3(W.foot_world[code]) - hp, m.foot_radius[code], fmaxf(m.boxmix[cls][code].margin, 0.0f) + 1e-4f, d, n)) {
          dist = d; nrm = n; pt = ld3(W.foot_world[code]) - (m.foot_radius[code] + 0.5f * d) * n;
        }
      }
    } else { dist = W.u2.c.lg_dist[code - 4]; nrm = ld3(GQ_BX_LGNRM(W) + 3 * (code - 4)); pt = ld3(W.u2.c.lg_pt[code - 4]); }
  }
  return true;
}

/* running totals of the contact list while world geoms are appended to it */
struct WorldAppend { int ncon, rows, invalid, reserve, ft /* bit k: foot k's calf body touches a world geom */, nself, ndrop /* contacts the capacity cut */; };

/* append the contacts of the collision items (lane = position in con_order: dist / nrm / pt) with one world geom of
 * contact-parameter class cls; rows / row budget as in the floor pass.  No barrier inside. */
template <bool CONE, bool PRIM = true>
__device__ inline void append_world_contacts(WaveMem& W, const GQ_MODEL GqDevModel& m, int cls, float mu_env, const PairHit& H, WorldAppend& S) {
  constexpr int NP = PRIM ? 4 : 1; /* points per item and world geom: only the exact pair routines return more than one */
  const int lane = lane_id();
  int& ncon = S.ncon; int& rows = S.rows; int& invalid = S.invalid; int& reserve = S.reserve; int& ft = S.ft;
  bool calf = false;
  int code = 0, body = 0, dim = 3, cnt = 0;
  float mu = 0.0f;
  bool tk[4] = {false, false, false, false};
  if (lane < 4 + m.nlg) {
    code = m.con_order[lane];
    const GQ_MODEL GqDevMix& X = m.boxmix[cls][code];
#pragma unroll
    for (int k = 0; k < NP; k++) { tk[k] = k < H.n && H.dist[k] < X.margin; cnt += tk[k] ? 1 : 0; }
    dim = X.dim;
    const float ff = m.boxcls_friction[cls][0]; /* _set_ground_friction leaves unnamed world boxes alone (quirk B8) */
    float fg;
    if (code < 4) { body = 3 + 3 * m.foot_leg[code]; calf = true; fg = mu_env >= 0.0f ? mu_env : m.foot_friction[code][0]; }
    else { const GQ_MODEL GqDevGeom& G = m.lg[code - 4]; body = G.body; calf = G.body > 0 && (G.body - 1) % 3 == 2; fg = G.friction[0]; }
    mu = fmaxf(1e-5f, X.rule == 0 ? fmaxf(ff, fg) : (X.rule == 1 ? ff : fg));
  }
  const bool touching = cnt > 0;
  if (ballot(touching) == 0) return;
  invalid |= ballot(touching && !calf) != 0;
#pragma unroll
  for (int k = 0; k < 4; k++) ft |= (ballot(touching && body == 3 + 3 * m.foot_leg[k]) != 0 ? 1 : 0) << k;
  /* ranks and rows: one prefix sum over (contacts, rows, reserved virtual rows), as in the floor pass */
  const int need = dim == 1 ? 1 : (CONE ? dim : 2 * (dim - 1));
  const int vres = (CONE && need > 1) ? need - 1 : 0;
  int idx0, rows0, res0, incl_all = 0;
  if constexpr (NP == 1) { /* one point per item: ballots and population counts (scalar unit) instead of the lane scan */
    const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    idx0 = ncon + popc64(ballot(touching) & lt);
    const bool kept = touching && idx0 < GQ_MAXCON;
    const uint64_t m1 = ballot(kept && need == 1), m3 = ballot(kept && need == 3), m4 = ballot(kept && need == 4), m6 = ballot(kept && need == 6);
    rows0 = rows + popc64(m1 & lt) + 3 * popc64(m3 & lt) + 4 * popc64(m4 & lt) + 6 * popc64(m6 & lt);
    res0 = CONE ? reserve + 2 * popc64(m3 & lt) + 5 * popc64(m6 & lt) : 0;
  } else {
    const int packed = cnt | ((cnt * need) << 8) | ((cnt * vres) << 18);
    incl_all = wave_incl_scan(packed);
    const int excl = incl_all - packed;
    idx0 = ncon + (excl & 0xff); rows0 = rows + ((excl >> 8) & 0x3ff); res0 = reserve + ((excl >> 18) & 0x3ff);
  }
  int nfit = 0, j = 0;
#pragma unroll
  for (int k = 0; k < NP; k++) {
    if (tk[k]) {
      const int idx = idx0 + j, row0 = rows0 + j * need, res = res0 + (j + 1) * vres;
      /* the list is a PREFIX of MuJoCo's: once a contact has been cut (S.ndrop, wave-uniform), no later one is taken even if it is small enough */
      const bool fits = S.ndrop == 0 && idx < GQ_MAXCON && row0 + need + res <= (CONE ? 64 : GQ_MAXEFC) && row0 + need <= GQ_MAXEFC;
      if (fits) {
        const GQ_MODEL GqDevMix& X = m.boxmix[cls][code];
        W.con_geom[idx] = code; W.con_body[idx] = body; W.con_dim[idx] = dim; W.con_row[idx] = row0;
        W.con_dist[idx] = H.dist[k]; W.con_inc[idx] = X.includemargin; W.con_mu[idx] = mu;
        st3(W.con_pos[idx], H.pos[k]);
        W.con_solref[idx][0] = X.solref[0]; W.con_solref[idx][1] = X.solref[1];
#pragma unroll
        for (int q = 0; q < 5; q++) W.con_solimp[idx][q] = X.solimp[q];
        st3(GQ_BX_CONNRM(W) + 3 * idx, hit_nrm(H, k));
        GQ_BX_WCLS(W)[idx] = cls;
        nfit++;
      }
      j++;
    }
  }
  if constexpr (NP == 1) {
    const uint64_t f1 = ballot(nfit && need == 1), f3 = ballot(nfit && need == 3), f4 = ballot(nfit && need == 4), f6 = ballot(nfit && need == 6);
    ncon += popc64(f1 | f3 | f4 | f6);
    rows += popc64(f1) + 3 * popc64(f3) + 4 * popc64(f4) + 6 * popc64(f6);
    if constexpr (CONE) reserve += 2 * popc64(f3) + 5 * popc64(f6);
    S.ndrop += popc64(ballot(touching)) - popc64(f1 | f3 | f4 | f6);
  } else {
    const int tot = bcast(wave_incl_scan(nfit | ((nfit * need) << 8) | ((nfit * vres) << 18)), 63);
    const int found = bcast(incl_all, 63) & 0xff; /* every touching point of this world geom */
    ncon += tot & 0xff;
    rows += (tot >> 8) & 0x3ff;
    if constexpr (CONE) reserve += (tot >> 18) & 0x3ff;
    S.ndrop += found - (tot & 0xff);
  }
}

/* closest points of the segments p1 + s d1 and p2 + t d2, s, t in [0, 1] (Ericson, Real-Time Collision Detection 5.1.9;
 * restated identically in oracle/gq_oracle.c::closest_seg_seg); degenerate segments (spheres) included */
__device__ __forceinline__ void closest_seg_seg(V3 p1, V3 q1, V3 p2, V3 q2, V3& c1, V3& c2) {
  const V3 d1 = q1 - p1, d2 = q2 - p2, r = p1 - p2;
  const float a = dot(d1, d1), e = dot(d2, d2), f = dot(d2, r), EPS = 1e-12f;
  float s, t;
  if (a <= EPS && e <= EPS) { s = 0.0f; t = 0.0f; }
  else if (a <= EPS) { s = 0.0f; t = med3(fdiv(f, e), 0.0f, 1.0f); }
  else {
    const float c = dot(d1, r);
    if (e <= EPS) { t = 0.0f; s = med3(fdiv(-c, a), 0.0f, 1.0f); }
    else {
      const float b = dot(d1, d2), den = a * e - b * b;
      s = den > 1e-6f * a * e ? med3(fdiv(b * f - c * e, den), 0.0f, 1.0f) : 0.0f; /* (nearly) parallel: any s will do; 0 */
      const float ie = fast_rcp(e), ia = fast_rcp(a);
      t = (b * s + f) * ie;
      if (t < 0.0f) { t = 0.0f; s = med3(-c * ia, 0.0f, 1.0f); }
      else if (t > 1.0f) { t = 1.0f; s = med3((b - c) * ia, 0.0f, 1.0f); }
    }
  }
  c1 = p1 + s * d1; c2 = p2 + t * d2;
}

/* contact frame word of a robot-robot contact: con_body = body2 | (body1 + 1) << 8, con_geom = item2 | (item1 + 1) << 8
 * (the high byte is 0 for contacts with a world geom); world class -2 */
#define GQ_CON_BODY2(x) ((x) & 0xff)
#define GQ_CON_BODY1(x) (((x) >> 8) - 1)   /* -1: world */
#define GQ_WCLS_SELF (-2)

/* S6, robot self-collision (mj_collision between two bodies of the robot; gym_quadruped_amd/selfcol.py has the pair
 * filter and the capsule proxies).  The proxies' end points are taken to the world once (lane = collision item); then
 * lane = geom pair, 64 pairs per pass: closest points of the two capsule axes, one contact per pair, normal from geom1 to
 * geom2, point midway between the surfaces.  Models with more than 128 pairs first run a body-pair broad phase (bounding
 * spheres of the bodies' proxies) and skip the passes whose pairs all belong to far-apart bodies.  Contacts are
 * appended after the world contacts in (body pair, geom1, geom2) order - MuJoCo's order - under the same row budget. */
/* model words of the self-collision stage that do not depend on the state: fetched at the start of S6, so that their
 * memory latency is spent under the floor scan instead of in front of the pair test */
struct SelfPrefetch { float caps[7], bsph[4]; int32_t body, it1[2], it2[2], bp[2]; };
__device__ __forceinline__ SelfPrefetch self_prefetch(const GQ_MODEL GqDevModel& m, const int nlg, const int nsp) {
  const int lane = lane_id();
  SelfPrefetch P;
  const int it = lane < 4 + nlg ? lane : 0;
#pragma unroll
  for (int i = 0; i < 7; i++) P.caps[i] = m.item_caps[it][i];
#pragma unroll
  for (int i = 0; i < 4; i++) P.bsph[i] = m.item_bsph[it][i];
  P.body = m.item_body[it];
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const int p = h * GQ_WAVE + lane < nsp ? h * GQ_WAVE + lane : 0;
    P.it1[h] = m.sp[p].it1; P.it2[h] = m.sp[p].it2; P.bp[h] = m.sp[p].bp;
  }
  return P;
}

/* a collision item as a shape of the convex routine (wave-uniform): feet and sphere / capsule geoms are their cores (the world end points of
 * the self-collision table + radius), everything else is its vertex cloud in the geom's frame */
__device__ inline void self_item_shape(const WaveMem& W, const GQ_MODEL GqDevModel& m, const int it, const float* k, LdsF dst) {
  CvxShape S;
  const int pt = it < 4 ? 2 : m.lg[it - 4].ptype;
  if (pt == 2 || pt == 3) {
    S.kind = 2; S.adr = 0; S.num = 2; S.pm = -1; S.t = ld3(k); S.h = ld3(k + 3); S.r = k[6];
#pragma unroll
    for (int i = 0; i < 9; i++) S.R[i] = (i % 4 == 0) ? 1.0f : 0.0f;
  } else {
    const GQ_MODEL GqDevGeom& G = m.lg[it - 4];
    const float* Rb = W.xmat[G.body];
    S.kind = 0; S.adr = G.plane_adr; S.num = G.cloud_num; S.pm = G.cap_adr; S.r = G.radius; S.h = v3(0.0f, 0.0f, 0.0f);
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) S.R[3 * i + j] = Rb[3 * i] * G.mat[j] + Rb[3 * i + 1] * G.mat[3 + j] + Rb[3 * i + 2] * G.mat[6 + j];
    S.t = ld3(W.xpos[G.body]) + matvec(Rb, ld3(G.pos));
  }
  cvx_shape_store(dst, S);
}
/* ... and its oriented bounding box for the mid phase (ONE lane): the cloud's geom-frame box grown by the radius; a core's world-axis box */
__device__ inline void self_item_obb(const WaveMem& W, const GQ_MODEL GqDevModel& m, const int it, const float* k, V3& c, float* R, V3& h) {
  const int pt = it < 4 ? 2 : m.lg[it - 4].ptype;
  if (pt == 2 || pt == 3) {
    const V3 p0 = ld3(k), p1 = ld3(k + 3);
    c = 0.5f * (p0 + p1);
    h = v3(0.5f * fabsf(p1.x - p0.x) + k[6], 0.5f * fabsf(p1.y - p0.y) + k[6], 0.5f * fabsf(p1.z - p0.z) + k[6]);
#pragma unroll
    for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1.0f : 0.0f;
  } else {
    const GQ_MODEL GqDevGeom& G = m.lg[it - 4];
    const float* Rb = W.xmat[G.body];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) R[3 * i + j] = Rb[3 * i] * G.mat[j] + Rb[3 * i + 1] * G.mat[3 + j] + Rb[3 * i + 2] * G.mat[6 + j];
    c = ld3(W.xpos[G.body]) + matvec(Rb, ld3(G.pos)) + matvec(R, ld3(G.aabb_c));
    h = v3(G.aabb_h[0] + G.radius, G.aabb_h[1] + G.radius, G.aabb_h[2] + G.radius);
  }
}

/* (a macro: __builtin_expect has to stand in the function that branches on it - the hint of an inlined helper is lowered away before the inlining) */
#define GQ_COLD_HINT(c) (COLD ? __builtin_expect((long)(c), 0L) : (long)(c)) /* (the front end folds the constant arm away) */

/* the direction along which a pair was found apart -> its row of the axis cache: unit length, base frame (ONE lane; 0 0 0: nothing to remember) */
__device__ inline void sepc_store(const WaveMem& W, float* row, V3 d) {
  const float dd = dot(d, d);
  V3 b = v3(0.0f, 0.0f, 0.0f);
  if (dd > 1e-24f) b = fast_rsqrt(dd) * matTvec(W.xmat[0], d);
  row[0] = b.x; row[1] = b.y; row[2] = b.z;
}

/* third mid-phase test of a convex self pair, ONE lane: the two shapes along the line of their origins (the direction GJK tries first) - a
 * hull by the upper bound of its support function (cvx_hgrid), a sphere / capsule core exactly - or along `dir` (the axis cache: the direction
 * the pair was found apart along a step ago).  true: farther apart than margin + radii,
 * the pair needs no support query (three in four of the pairs that pass the oriented boxes end at the routine's first one). */
__device__ inline bool self_hulls_apart(const WaveMem& W, const GQ_MODEL GqDevModel& m, const GQ_MODEL float* vx, int it1, const float* k1, const float* R1,
                                        int it2, const float* k2, const float* R2, float marg, bool use_dir = false, V3 dir = {0.0f, 0.0f, 0.0f}) {
  V3 t[2]; float r[2]; int adr[2]; bool seg[2];
#pragma unroll
  for (int s = 0; s < 2; s++) {
    const int it = s ? it2 : it1; const float* k = s ? k2 : k1;
    const int pt = it < 4 ? 2 : m.lg[it - 4].ptype;
    seg[s] = pt == 2 || pt == 3;
    if (seg[s]) { t[s] = 0.5f * (ld3(k) + ld3(k + 3)); r[s] = k[6]; adr[s] = -1; }
    else { const GQ_MODEL GqDevGeom& G = m.lg[it - 4]; t[s] = ld3(W.xpos[G.body]) + matvec(W.xmat[G.body], ld3(G.pos)); r[s] = G.radius; adr[s] = G.hgrid_adr; if (adr[s] < 0) return false; }
  }
  const V3 d = use_dir ? dir : t[1] - t[0];
  const float dd = dot(d, d);
  if (!(dd > 1e-12f)) return false;
  const float hA = seg[0] ? fmaxf(dot(ld3(k1), d), dot(ld3(k1 + 3), d)) : dot(t[0], d) + cvx_hgrid(vx, adr[0], matTvec(R1, d));                 /* max over A of v . d */
  const float hB = seg[1] ? fmaxf(-dot(ld3(k2), d), -dot(ld3(k2 + 3), d)) : -dot(t[1], d) + cvx_hgrid(vx, adr[1], matTvec(R2, -1.0f * d));       /* max over B of v . (-d) */
  return -hA - hB > (marg + r[0] + r[1] + 1e-5f) * fast_sqrt(dd);
}

/* COLD: tell the register allocator that the convex block is rarely entered (see there) - the world-box variants, whose robots mostly have no hull
 * pair (aliengo perlin + 8 %), do; the flat-scene variants, whose launch on the headline workload IS the convex routine, do not (- 4 % with it) */
template <bool CONE, bool PRIM = true, bool COLD = false>
__device__ inline void append_self_contacts(WaveMem& W, const GQ_MODEL GqDevModel& m, float mu_env, WorldAppend& S, const SelfPrefetch& pre, const StepConsts& K, const int nlg, const GQ_MODEL GqDevBatch& Bt, float* xdbg = nullptr, const int env = 0) {
  constexpr int NP = PRIM ? 4 : 1; /* points per pair: only the exact pair routines return more than one */
  const int lane = lane_id();
  const int nsp = K.nsp;
  if (nsp == 0) return;
#ifdef GQ_XQ_OFF /* experiment builds: the pair exchange compiled out */
  int32_t* const xq_tab = nullptr; const int xq_slots = 0; const int xq_pre = 0; (void)Bt;
#else
  int32_t* const xq_tab = Bt.xq; const int xq_slots = Bt.xq_slots; /* (fetched here: the scalar loads return behind the end points and the pair cull) */
  /* the first half of this wavefront's window of the table - its word 31: when a pair was last published into the window, its word 63: when
   * an env without convex work (a potential helper) last passed by - fetched here, so that the latency passes behind the end points and the cull */
  float* const sepc = Bt.sepc ? Bt.sepc + (size_t)env * Bt.sepc_stride : nullptr; /* this env's rows of the separating-axis cache */
  int xq_pre = 0;
  if (xq_tab) { Xq X0; X0.q = xq_tab; X0.slots = xq_slots; xq_pre = ld_pub(xq_tab + xq_window(X0, wave_index()) + lane); }
#endif
  const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  /* world end points of every item's proxy capsule, once: lane = collision item; scratch in the J block, which is free
   * until S7 (the spatial-dynamics scratch it overlays is dead since S5) */
  /* per item: end points p0, p1, radius, and the bounding sphere (centre, radius) the first pass tests */
  float(*cw)[12] = reinterpret_cast<float(*)[12]>(&W.u.B[0][0]);
  { /* lane = item; lanes past the last item repeat item 0 (self_prefetch fetched its record for them): no exec-mask block */
    const int it = lane < 4 + nlg ? lane : 0;
    const int b = pre.body;
    const V3 o = ld3(W.xpos[b]);
    const V3 e0 = o + matvec(W.xmat[b], v3(pre.caps[0], pre.caps[1], pre.caps[2])), e1 = o + matvec(W.xmat[b], v3(pre.caps[3], pre.caps[4], pre.caps[5]));
    st3(cw[it], e0); st3(cw[it] + 3, e1);
    cw[it][6] = pre.caps[6];
    st3(cw[it] + 8, o + matvec(W.xmat[b], v3(pre.bsph[0], pre.bsph[1], pre.bsph[2]))); /* broad-phase sphere: the primitive itself where the item is one */
    cw[it][11] = pre.bsph[3];
  }
  /* models with many pairs: body-pair broad phase, so that passes whose pairs all belong to far-apart bodies are skipped */
  uint64_t near[2] = {~0ull, ~0ull};
  if (nsp > 2 * GQ_WAVE) {
#pragma unroll
    for (int half = 0; half < 2; half++) {
      const int p = half * GQ_WAVE + lane;
      bool nr = false;
      if (p < m.nbp) {
        const GQ_MODEL GqDevBodyPair& P = m.bp[p];
        const V3 d = ld3(W.xpos[P.b2]) + matvec(W.xmat[P.b2], ld3(m.body_sph[P.b2])) - ld3(W.xpos[P.b1]) - matvec(W.xmat[P.b1], ld3(m.body_sph[P.b1]));
        const float rr = m.body_sph[P.b1][3] + m.body_sph[P.b2][3] + K.self_margin;
        nr = dot(d, d) < rr * rr;
      }
      near[half] = ballot(nr);
    }
  }
  /* lane = pass: does any body pair of the pass's 64 geom pairs come near?  (one load per lane, in flight with the spheres above) */
  /* (elliptic variants only - go2 372 pairs +0.7 %, go1 655 pairs +0.8 %; in the pyramidal ones the test cost what it saved: b2 -0.2 %,
   * and the headline's code, 90 pairs and never a skipped pass, moved by -0.3 % with it) */
  uint64_t live_pass = ~0ull;
  if constexpr (CONE) if (nsp > 2 * GQ_WAVE) {
    const int np = (nsp + GQ_WAVE - 1) / GQ_WAVE, lp = lane < np ? lane : 0;
    const uint64_t pb0 = m.sp_pass_bp[lp][0], pb1 = m.sp_pass_bp[lp][1];
    live_pass = ballot(lane < np && ((near[0] & pb0) | (near[1] & pb1)) != 0);
  }
  if (K.self_cut == 1) { wave_barrier(); return; }
  wave_barrier();
  GQ_SUB(W, 1, 10); /* proxy end points */
  /* pass A, lane = geom pair, 64 pairs per pass: bounding spheres of the two capsules.  The survivors' pair indices are
   * compacted into a list (scratch behind the end points, pair order kept), so that the closest-point test runs once over
   * the candidates instead of once per pass - and not at all in the many poses where no pair comes close */
  int32_t* list = reinterpret_cast<int32_t*>(&W.u.B[30][0]);
  int ncand = 0;
#pragma unroll 1
  for (int p0 = 0; p0 < nsp; p0 += GQ_WAVE) {
    if constexpr (CONE) if (!((live_pass >> (p0 / GQ_WAVE)) & 1ull)) continue; /* wave-uniform: every body pair of this pass is far apart */
    const int p = p0 + lane;
    bool cand;
    { /* branch-free: lanes past the last pair test pair 0 (the prefetch's fall-back) and are masked */
      int it1, it2, bp; /* the first two passes come prefetched */
      if (p0 == 0) { it1 = pre.it1[0]; it2 = pre.it2[0]; bp = pre.bp[0]; }
      else if (p0 == GQ_WAVE) { it1 = pre.it1[1]; it2 = pre.it2[1]; bp = pre.bp[1]; }
      else { const GQ_MODEL GqDevSelfPair& P = m.sp[p < nsp ? p : 0]; it1 = P.it1; it2 = P.it2; bp = P.bp; }
      const bool nr = (near[(bp >> 6) & 1] >> (bp & 63)) & 1;
      const float* k1 = cw[it1] + 8;
      const float* k2 = cw[it2] + 8;
      const V3 dm = ld3(k2) - ld3(k1);
      const float reach = k1[3] + k2[3] + K.self_margin;
      cand = p < nsp && nr && dot(dm, dm) < reach * reach;
    }
    const uint64_t cm = ballot(cand);
    if (cm == 0) continue;
    const int at = ncand + popc64(cm & lt);
    if (cand && at < 2 * GQ_WAVE) list[at] = p;
    ncand += popc64(cm);
  }
  GQ_SUB(W, 1, 11); /* pair cull */
  const int xq_act = bcast(xq_pre, 31), xq_hlp = bcast(xq_pre, 63);
  if (K.self_cut == 2) return;
  if (ncand == 0) { /* nothing of its own to do here: leave word of that, and stay only if pairs have come this way lately */
    if (xq_tab == nullptr) return;
    Xq X0; X0.q = xq_tab; X0.slots = xq_slots;
    xq_mark_helper(X0, wave_index(), xq_hlp);
    if (!xq_is_hot(xq_act, wall_clock64())) return;
  }
  if (ncand > 2 * GQ_WAVE) ncand = 2 * GQ_WAVE; /* more than 128 close pairs: the robot is a knot; the row budget is long spent */
  wave_barrier();
  const int npass = ncand > 0 ? ncand : 1; /* (a batch with a pair exchange: an env without a candidate still passes by the convex block once - it may have time for others, gq_exchange.h) */
#pragma unroll 1
  for (int c0 = 0; c0 < npass; c0 += GQ_WAVE) { /* pass B, lane = candidate pair */
    const bool cand = c0 + lane < ncand;
    const int p = cand ? list[c0 + lane] : 0;
    PairHit H;
    H.n = 0;
    int it1 = 0, it2 = 0;
    bool cvx = false;
    if (cand) {
      const GQ_MODEL GqDevSelfPair& Pp = m.sp[p];
      it1 = Pp.it1; it2 = Pp.it2;
      const int kind = Pp.kind;
      const float marg = Pp.mix.margin;
      const float* k1 = cw[it1];
      const float* k2 = cw[it2];
      if (kind == 0) { /* capsule proxies: closest points of the two axes */
        V3 c1, c2;
        closest_seg_seg(ld3(k1), ld3(k1 + 3), ld3(k2), ld3(k2 + 3), c1, c2);
        const V3 d = c2 - c1;
        const float l2 = dot(d, d), len = fast_sqrt(l2), dist = len - k1[6] - k2[6];
        if (dist < marg && len >= 1e-9f) {
          const V3 nrm = fast_rcp(len) * d;
          H.n = 1; H.dist[0] = dist; H.nrm[0] = nrm; H.pos[0] = c1 + (k1[6] + 0.5f * dist) * nrm;
        }
      } else if (kind == 4) { /* a hull / cylinder is involved: the convex routine, one pair at a time below - here its mid phase, the two shapes' oriented boxes */
        V3 c1, h1, c2, h2; float R1[9], R2[9];
        self_item_obb(W, m, it1, k1, c1, R1, h1);
        self_item_obb(W, m, it2, k2, c2, R2, h2);
        cvx = !obb_apart(c1, R1, h1, c2, R2, h2, marg);
        if (cvx) cvx = !self_hulls_apart(W, m, K.vx, it1, k1, R1, it2, k2, R2, marg); /* the hulls themselves, by their support grids */
        if (cvx && sepc) { /* ... and along the direction that told them apart a step ago (two links a few centimetres apart for ever: hyqreal1's trunk and upper legs) */
          const float* c = sepc + 3 * Pp.cidx;
          const V3 dw = matvec(W.xmat[0], v3(c[0], c[1], c[2]));
          if (dot(dw, dw) > 0.25f) cvx = !self_hulls_apart(W, m, K.vx, it1, k1, R1, it2, k2, R2, marg, true, dw);
        }
      } else if constexpr (PRIM) { /* a box is involved: exact routines (gq_pairs.h); the box of kind 1 / 3 is item 1, of kind 2 item 2 */
        const int ib = kind == 2 ? it2 : it1;
        bool continue_pair = true;
        const GQ_MODEL GqDevGeom& G = m.lg[ib - 4];
        const float* Rb = W.xmat[G.body];
        float A[9];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
          for (int j = 0; j < 3; j++) A[3 * i + j] = Rb[3 * i] * G.mat[j] + Rb[3 * i + 1] * G.mat[3 + j] + Rb[3 * i + 2] * G.mat[6 + j];
        const V3 ca = ld3(W.xpos[G.body]) + matvec(Rb, ld3(G.pos)), ha = ld3(G.psize);
        if (kind == 3) {
          const GQ_MODEL GqDevGeom& G2 = m.lg[it2 - 4];
          { /* box 2's bounding sphere against box 1 itself (long thin link boxes have loose spheres): most candidates end here */
            V3 nn;
            if (point_box(matTvec(A, ld3(k2 + 8) - ca), ha, nn) - k2[11] >= marg) continue_pair = false;
          }
          if (continue_pair) {
          const float* Rb2 = W.xmat[G2.body];
          float A2[9];
#pragma unroll
          for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) A2[3 * i + j] = Rb2[3 * i] * G2.mat[j] + Rb2[3 * i + 1] * G2.mat[3 + j] + Rb2[3 * i + 2] * G2.mat[6 + j];
          box_box(ca, A, ha, ld3(W.xpos[G2.body]) + matvec(Rb2, ld3(G2.pos)), A2, ld3(G2.psize), marg, H);
          }
        } else {
          const float* kc = kind == 1 ? k2 : k1; /* the sphere / capsule: its proxy is the geom itself */
          capsule_box(ld3(kc), ld3(kc + 3), kc[6], ca, A, ha, marg, H);
          if (kind == 2) { /* normal from the box (item 2) to the capsule (item 1): turn it to run from item 1 to item 2 */
#pragma unroll
            for (int k = 0; k < 2; k++) H.nrm[k] = -1.0f * H.nrm[k];
          }
        }
      }
    }
    { /* the convex pairs that passed their mid phase, one after the other on the whole wavefront (gq_convex.h) - or, when the batch has a
       * pair exchange and this env several pairs, shared with wavefronts that have time (gq_exchange.h).  ONE call site of the routine
       * serves the pairs computed here and the published pairs taken back. */
      uint64_t cm = ballot(cvx);
      if (K.self_cut == 5) cm = 0;        /* profiling aid: the mid phase without the convex routine */
      if (K.self_cut == 6) cm &= cm - 1;  /* ... and without the first pair that reaches it */
      Xq X; X.q = xq_tab; X.slots = xq_slots;
      uint64_t local = cm, own = 0; /* pairs to run here / published */
      int myslot = -1;              /* lane = pair: the slot it reserved */
      /* an env without convex work of its own has 40 us to spare before the launch's entangled envs are through: it lingers here for a few
       * microseconds - if its group of the table has seen pairs lately - and takes what gets published meanwhile */
      bool linger = X.q != nullptr && cm == 0 && c0 + GQ_WAVE >= npass && xq_is_hot(xq_act, wall_clock64());
      int t_ref = 0; bool t_set = false; /* start of the wait in progress, 100 MHz ticks (low word): an owner's deadline or the lingering */
#define GQ_XSTAT(i, v) do { if (xdbg && lane == 0) xdbg[i] = (float)(v); } while (0)
#define GQ_XTIME(i) GQ_XSTAT(i, wall_clock64() & 0xFFFFF)
      int x_back = 0, x_help = 0; long long x_ticks = 0, x_t0 = 0; bool x_ld = false;
      if (cm) GQ_XSTAT(0, popc64(cm));
      if (X.q != nullptr && cm == 0 && ncand != 0 && c0 + GQ_WAVE >= npass) xq_mark_helper(X, wave_index(), xq_hlp);
      if (GQ_COLD_HINT(X.q != nullptr && (cm & (cm - 1)) != 0 && xq_is_hot(xq_hlp, wall_clock64()))) { /* two pairs or more - and envs with time on their hands around (hyqreal1 on boxes: every env has four or five pairs, publishing would be pure overhead): keep the first, publish the others - at once, helpers come by only so often */
        const uint64_t rest = cm & (cm - 1);
        if ((rest >> lane) & 1ull) myslot = xq_reserve(X, wave_index(), lane);
        own = ballot(myslot >= 0);
        local = cm & ~own; /* (no free slot among a pair's candidates: it stays here) */
        for (uint64_t r = own; r;) {
          const int jj = ffs64(r); r &= r - 1;
          const int pj = bcast(p, jj), i1 = bcast(it1, jj), i2 = bcast(it2, jj), sj = bcast(myslot, jj);
          wave_barrier();
          self_item_shape(W, m, i1, cw[i1], GQ_CVX_SHP(W));
          self_item_shape(W, m, i2, cw[i2], GQ_CVX_SHP(W) + GQ_CVX_SHAPE_WORDS);
          wave_barrier();
          xq_put(X, sj, GQ_CVX_SHP(W), m.sp[pj].mix.margin);
        }
        publish_fence();
        if (myslot >= 0) { xq_ready(X, myslot); xq_mark_active(X, myslot); }
        GQ_XSTAT(1, popc64(own)); GQ_XTIME(2);
      }
      /* (marked unlikely for the register allocator's sake: block frequencies are its spill weights, and the routine's loops otherwise outweigh
       * values that live across the whole step - they were spilled all over the kernel, 57 scratch instructions in gq_step_body.h alone) */
      if (GQ_COLD_HINT(local != 0 || own != 0 || linger))
#pragma unroll 1
      for (;;) { /* wave-uniform */
        int j = -1, slot = -1;
        if (local) { j = ffs64(local); local &= local - 1; }
        else if (own) { /* the pairs kept here are done: watch the published ones */
          if (xdbg && !x_ld) { GQ_XTIME(3); x_ld = true; }
          const int s = myslot >= 0 ? ld_pub(xq_state(X, myslot)) : XQ_DONE;
          const uint64_t ready = ballot(s == XQ_READY), busy = ballot(s == XQ_CLAIMED);
          if (ready) { /* nobody has taken it yet: take it back */
            j = ffs64(ready);
            slot = bcast(myslot, j);
            if (!xq_claim(X, slot)) continue; /* a helper was faster */
            j = -1; x_back++;
          } else if (busy) {
            if (!t_set) { t_ref = (int)wall_clock64(); t_set = true; }
            if ((int)wall_clock64() - t_ref < GQ_XQ_OWNER_TICKS) { nap(); continue; }
            /* (never seen) a helper holds the pair for milliseconds: compute it here; its slot is left behind, results are not read from it */
            j = ffs64(busy);
            own &= ~(1ull << j);
            if (lane == j) myslot = -1;
          } else break; /* all DONE */
        } else if (linger) {
          if (!t_set) { t_ref = (int)wall_clock64(); t_set = true; }
          slot = xq_scan(X, wave_index());
          if (slot >= 0) {
            if (!xq_claim(X, slot)) continue;
            if (xdbg) { if (x_help == 0) GQ_XTIME(7); x_help++; x_t0 = wall_clock64(); }
          } else {
            if ((int)wall_clock64() - t_ref > GQ_XQ_LINGER_TICKS) break;
            nap();
            continue;
          }
        } else break;
        float marg;
        wave_barrier();
        if (j >= 0) {
          const int pj = bcast(p, j), i1 = bcast(it1, j), i2 = bcast(it2, j);
          self_item_shape(W, m, i1, cw[i1], GQ_CVX_SHP(W));
          self_item_shape(W, m, i2, cw[i2], GQ_CVX_SHP(W) + GQ_CVX_SHAPE_WORDS);
          marg = m.sp[pj].mix.margin;
        } else xq_get(X, slot, GQ_CVX_SHP(W), marg);
        wave_barrier();
        const bool hit = cvx_pair_wave(GQ_CVX_SHP(W), GQ_CVX_POLY_SELF(W), K.vx, K.vy, K.vz, marg);
        LdsCF out = GQ_CVX_SHP(W) + 2 * GQ_CVX_SHAPE_WORDS;
        if (j >= 0) {
          if (hit && lane == j) { H.n = 1; H.dist[0] = out[0]; H.nrm[0] = ld3(out + 1); H.pos[0] = ld3(out + 4); }
          if (!hit && lane == j && sepc) sepc_store(W, sepc + 3 * m.sp[p].cidx, ld3(out + 1));
        } else {
          wave_barrier(); xq_done(X, slot, hit, out);
          if (xdbg && linger) x_ticks += wall_clock64() - x_t0;
          if (linger && (int)wall_clock64() - t_ref > GQ_XQ_LINGER_TICKS) linger = false;
        }
      }
      if (xdbg) { if (own) { GQ_XTIME(4); GQ_XSTAT(5, x_back); } if (x_help) { GQ_XSTAT(6, x_help); GQ_XSTAT(8, x_ticks); } if (cm == 0 && X.q) GQ_XTIME(9); }
      if (own) { /* collect, and give the slots back */
        if (myslot >= 0) {
          const int32_t* it = xq_item(X, myslot);
          if (ld_pub(it + GQ_XQ_RES) != 0) {
            const float* r = reinterpret_cast<const float*>(it) + GQ_XQ_RES;
            H.n = 1; H.dist[0] = ld_pub(r + 1); H.nrm[0] = v3(ld_pub(r + 2), ld_pub(r + 3), ld_pub(r + 4)); H.pos[0] = v3(ld_pub(r + 5), ld_pub(r + 6), ld_pub(r + 7));
          }
          else if (sepc) { const float* r = reinterpret_cast<const float*>(it) + GQ_XQ_RES; sepc_store(W, sepc + 3 * m.sp[p].cidx, v3(ld_pub(r + 2), ld_pub(r + 3), ld_pub(r + 4))); }
          st_pub(xq_state(X, myslot), XQ_FREE); /* (the loads above have returned: H is used below) */
        }
      }
    }
    const int cnt = H.n;
    if (ballot(cnt > 0) == 0 || K.self_cut == 4) continue;
    /* (rare) append the touching pairs' points, in pair order */
    const GQ_MODEL GqDevSelfPair& P = m.sp[cnt > 0 ? p : 0];
    const int dim = P.mix.dim;
    float mu = 0.0f;
    if (cnt > 0) { /* sliding friction: _set_ground_friction rewrites the feet (quadruped_env.py:1277-1298) */
      const float f1 = it1 < 4 ? (mu_env >= 0.0f ? mu_env : m.foot_friction[it1][0]) : m.lg[it1 - 4].friction[0];
      const float f2 = it2 < 4 ? (mu_env >= 0.0f ? mu_env : m.foot_friction[it2][0]) : m.lg[it2 - 4].friction[0];
      mu = fmaxf(1e-5f, P.mix.rule == 0 ? fmaxf(f1, f2) : (P.mix.rule == 1 ? f1 : f2));
    }
    int& ncon = S.ncon; int& rows = S.rows; int& reserve = S.reserve;
    const int need = dim == 1 ? 1 : (CONE ? dim : 2 * (dim - 1));
    const int vres = (CONE && need > 1) ? need - 1 : 0;
    int idx0, rows0, res0, incl_all = 0;
    if constexpr (NP == 1) { /* one point per pair: ballots and population counts instead of the lane scan */
      idx0 = ncon + popc64(ballot(cnt > 0) & lt);
      const bool kept = cnt > 0 && idx0 < GQ_MAXCON;
      const uint64_t m1 = ballot(kept && need == 1), m3 = ballot(kept && need == 3), m4 = ballot(kept && need == 4), m6 = ballot(kept && need == 6);
      rows0 = rows + popc64(m1 & lt) + 3 * popc64(m3 & lt) + 4 * popc64(m4 & lt) + 6 * popc64(m6 & lt);
      res0 = CONE ? reserve + 2 * popc64(m3 & lt) + 5 * popc64(m6 & lt) : 0;
    } else {
      const int packed = cnt | ((cnt * need) << 8) | ((cnt * vres) << 18);
      incl_all = wave_incl_scan(packed);
      const int excl = incl_all - packed;
      idx0 = ncon + (excl & 0xff); rows0 = rows + ((excl >> 8) & 0x3ff); res0 = reserve + ((excl >> 18) & 0x3ff);
    }
    int nfit = 0;
#pragma unroll
    for (int k = 0; k < NP; k++) {
      if (k < cnt) {
        const int idx = idx0 + k, row0 = rows0 + k * need, res = res0 + (k + 1) * vres;
        const bool fits = S.ndrop == 0 && idx < GQ_MAXCON && row0 + need + res <= (CONE ? 64 : GQ_MAXEFC) && row0 + need <= GQ_MAXEFC;
        if (fits) {
          const int b1 = m.item_body[it1], b2 = m.item_body[it2];
          W.con_geom[idx] = it2 | ((it1 + 1) << 8) | (P.mix.rule << 16); W.con_body[idx] = b2 | ((b1 + 1) << 8); W.con_dim[idx] = dim; W.con_row[idx] = row0;
          W.con_dist[idx] = H.dist[k]; W.con_inc[idx] = P.mix.includemargin; W.con_mu[idx] = mu;
          st3(W.con_pos[idx], H.pos[k]);
          W.con_solref[idx][0] = P.mix.solref[0]; W.con_solref[idx][1] = P.mix.solref[1];
#pragma unroll
          for (int q = 0; q < 5; q++) W.con_solimp[idx][q] = P.mix.solimp[q];
          st3(GQ_BX_CONNRM(W) + 3 * idx, hit_nrm(H, k));
          GQ_BX_WCLS(W)[idx] = GQ_WCLS_SELF;
#ifdef GQ_EMU_TRACE
          if (getenv("GQ_EMU_TRACE")) printf("self contact idx %d items %d %d kind %d point %d dist %.6f pos %.5f %.5f %.5f nrm %.5f %.5f %.5f\n", idx, it1, it2, (int)m.sp[p].kind, k, (double)H.dist[k], (double)H.pos[k].x, (double)H.pos[k].y, (double)H.pos[k].z, (double)hit_nrm(H, k).x, (double)hit_nrm(H, k).y, (double)hit_nrm(H, k).z);
#endif
          nfit++;
        }
      }
    }
    if constexpr (NP == 1) {
      const uint64_t f1 = ballot(nfit && need == 1), f3 = ballot(nfit && need == 3), f4 = ballot(nfit && need == 4), f6 = ballot(nfit && need == 6);
      ncon += popc64(f1 | f3 | f4 | f6);
      rows += popc64(f1) + 3 * popc64(f3) + 4 * popc64(f4) + 6 * popc64(f6);
      if constexpr (CONE) reserve += 2 * popc64(f3) + 5 * popc64(f6);
      S.nself += popc64(f1 | f3 | f4 | f6);
      S.ndrop += popc64(ballot(cnt > 0)) - popc64(f1 | f3 | f4 | f6);
    } else {
      const int tot = bcast(wave_incl_scan(nfit | ((nfit * need) << 8) | ((nfit * vres) << 18)), 63);
      const int found = bcast(incl_all, 63) & 0xff;
      ncon += tot & 0xff;
      rows += (tot >> 8) & 0x3ff;
      if constexpr (CONE) reserve += (tot >> 18) & 0x3ff;
      S.nself += tot & 0xff;
      S.ndrop += found - (tot & 0xff);
    }
  }
}

/* S6 for a scene without world boxes / height field but with robot self-collision: general frames for the floor
 * contacts the floor pass left in W, then the robot-robot contacts.  Ends with a barrier. */
template <bool CONE>
__device__ __forceinline__ void stage_self_contacts(WaveMem& W, const GQ_MODEL GqDevModel& m, float mu_env, const SelfPrefetch& pre, const StepConsts& K, const int nlg, const GQ_MODEL GqDevBatch& Bt, float* xdbg = nullptr, const int env = 0) {
  const int lane = lane_id();
  WorldAppend S;
  S.ncon = uniform(W.ncon); S.rows = uniform(W.nefc); S.invalid = 0; S.reserve = 0; S.nself = 0; S.ndrop = uniform(W.ndrop);
  const int ncon = S.ncon;
  { /* floor contacts: normal z, world geom = floor - written for every slot of the list (contacts appended below overwrite theirs) */
    const int lc = lane < GQ_MAXCON ? lane : GQ_MAXCON - 1;
    st3(GQ_BX_CONNRM(W) + 3 * lc, v3(0.0f, 0.0f, 1.0f));
    GQ_BX_WCLS(W)[lc] = -1;
  }
  if constexpr (CONE)
    for (int c = 0; c < ncon; c++) { const int d = uniform(W.con_dim[c]); S.reserve += d > 1 ? d - 1 : 0; }
  append_self_contacts<CONE>(W, m, mu_env, S, pre, K, nlg, Bt, xdbg, env);
  { W.ncon = S.ncon; W.nefc = S.rows; W.nself = S.nself; W.ndrop = S.ndrop; } /* (every lane: the same words) */
  wave_barrier();
}

/* S6 (BOXES): append the contacts with the world boxes to the list the floor pass left in W (ncon, nefc, invalid,
 * foot_touch are updated; rows / row budget as in the floor pass).  Ends with a barrier. */
/* (forced inline: out of line - the inliner's choice on some variants - the item record and the prefetch block it takes by reference are
 * materialised in scratch memory, 200 bytes per lane) */
template <bool CONE, bool SELF, bool PRIM>
__device__ __forceinline__ void stage_box_contacts(WaveMem& W, const GQ_MODEL GqDevModel& m, const GQ_MODEL float* vx, const GQ_MODEL float* vy, const GQ_MODEL float* vz,
                                          double bx, double by, float mu_env, const SelfPrefetch& pre, const ItemRegs& IT, const StepConsts& K, const int nlg, const GQ_MODEL GqDevBatch& Bt, float* xdbg = nullptr, const int env = 0) {
  const int lane = lane_id();
  const PrimLane PL = prim_lane(W, m, IT, PRIM && lane < 4 + m.nlg, PRIM); /* all the box loop keeps of the item record */
  WorldAppend S;
  S.ncon = uniform(W.ncon); S.rows = uniform(W.nefc); S.invalid = uniform(W.invalid); S.reserve = 0; S.nself = 0; S.ndrop = uniform(W.ndrop);
  S.ft = uniform(W.foot_touch) & 15;
  const int ncon = S.ncon;
  { /* floor contacts: normal z, world geom = floor - written for every slot of the list (contacts appended below overwrite theirs) */
    const int lc = lane < GQ_MAXCON ? lane : GQ_MAXCON - 1;
    st3(GQ_BX_CONNRM(W) + 3 * lc, v3(0.0f, 0.0f, 1.0f));
    GQ_BX_WCLS(W)[lc] = -1;
  }
  if constexpr (CONE)
    for (int c = 0; c < ncon; c++) { const int d = uniform(W.con_dim[c]); S.reserve += d > 1 ? d - 1 : 0; }
  uint64_t cand[2];
  {
    V3 cg; float rg;
    item_sphere(W, m, false, cg, rg);
    box_candidates(W, m, bx, by, 0.0f, cand, cg, rg);
    if (lane < GQ_MAXLG) { st3(GQ_BX_ISPH(W) + 4 * lane, cg); GQ_BX_ISPH(W)[4 * lane + 3] = rg; }
    item_obb_store(W, m); /* the clouds' oriented boxes for box_item_scan's second bounding test */
  }
  wave_barrier();
#pragma unroll 1
  for (int half = 0; half < 2; half++) {
    uint64_t todo = cand[half];
    while (todo) { /* wave-uniform */
      const int b = half * GQ_WAVE + ffs64(todo);
      todo &= todo - 1;
      PairHit H;
      const float* sph = GQ_BX_ISPH(W) + 4 * opaque_lane(lane < GQ_MAXLG ? lane : 0);
      const V3 cg = ld3(sph); const float rg = sph[3];
      if (!box_item_scan<PRIM, 1, true>(W, m, vx, vy, vz, b, bx, by, 0.0f, cg, rg, PL, H)) continue;
      append_world_contacts<CONE, PRIM>(W, m, m.box[b].cls, mu_env, H, S);
      wave_barrier();
    }
  }
  if (m.hf_nrow > 0) { /* the scene's height field: one more world geom */
    float dist; V3 nrm, pt;
    const float* sph = GQ_BX_ISPH(W) + 4 * opaque_lane(lane < GQ_MAXLG ? lane : 0);
    const V3 cg = ld3(sph); const float rg = sph[3];
    if (hfield_item_scan(W, m, vx, vy, vz, bx, by, 0.0f, cg, rg, dist, nrm, pt)) {
      PairHit H;
      H.n = dist < 1e29f ? 1 : 0; H.dist[0] = dist; H.nrm[0] = nrm; H.pos[0] = pt;
      append_world_contacts<CONE, false>(W, m, m.hf_cls, mu_env, H, S);
      wave_barrier();
    }
  }
  if constexpr (SELF) {
    (void)pre;
    const SelfPrefetch pre_now = self_prefetch(m, nlg, K.nsp); /* not prefetched in front of the box loop: it would sit in registers (or scratch) across it */
    append_self_contacts<CONE, PRIM, true>(W, m, mu_env, S, pre_now, K, nlg, Bt, xdbg, env);
  }
  if (lane == 0) {
    W.ncon = S.ncon; W.nefc = S.rows; W.invalid = S.invalid; W.nself = S.nself; W.ndrop = S.ndrop;
    W.foot_touch = S.ft;
  }
  wave_barrier();
}

}  // namespace gq
