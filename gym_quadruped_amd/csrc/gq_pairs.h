/*
 * gq_pairs.h - exact narrow phases of primitive pairs, evaluated by ONE lane: sphere / capsule against box and box against
 * box (robot-robot contacts and robot geoms on the static world boxes).  MuJoCo runs mjc_SphereBox, mjc_CapsuleBox (<= 2 points)
 * and mjc_BoxBox (<= 8 points) there; what is computed here is their geometry - exact closest features, separating-axis
 * penetration - with a manifold of at most 2 / 4 points chosen by the rules below, restated line for line in
 * oracle/gq_oracle.c (capsule_box, box_box; pinned there against brute-force geometry, tests/test_oracle_invariants.py).
 * Conventions: box frames as (centre c, matrix R whose COLUMNS are the box axes, row-major float[9], half sizes h); every
 * routine returns the normal pointing from the (first) box to the other geom, the point midway between the two surfaces and
 * the signed distance.
 */
#pragma once
#include "gq_step_kernel.h"

namespace gq {

/* (two normals: the <= 2 points of a capsule have one each, the <= 4 of a box-box manifold share nrm[0] - six registers less across the
 * world-box loop of the step kernel) */
struct PairHit { int n; float dist[4]; V3 pos[4]; V3 nrm[2]; };
__device__ __forceinline__ V3 hit_nrm(const PairHit& H, const int k) { return k == 1 ? H.nrm[1] : H.nrm[0]; }

/* the lane's item record (GqDevModel::item), in registers */
struct ItemRegs {
  int code, body, dim, fric_rule, ptype, calf;
  float margin, inc, friction0, radius, solref[2], solimp[5], psize[3];
  V3 pos;
  float mat[9];
};
__device__ __forceinline__ ItemRegs item_fetch(const GQ_MODEL GqDevModel& m, const int it) {
  const GQ_MODEL GqDevItem& I = m.item[it];
  ItemRegs R;
  R.code = I.code; R.body = I.body; R.dim = I.dim; R.fric_rule = I.fric_rule; R.ptype = I.ptype; R.calf = I.calf;
  R.margin = I.margin; R.inc = I.inc; R.friction0 = I.friction0; R.radius = I.radius;
  R.solref[0] = I.solref[0]; R.solref[1] = I.solref[1];
#pragma unroll
  for (int q = 0; q < 5; q++) R.solimp[q] = I.solimp[q];
#pragma unroll
  for (int q = 0; q < 3; q++) R.psize[q] = I.psize[q];
  R.pos = ld3(I.pos);
#pragma unroll
  for (int q = 0; q < 9; q++) R.mat[q] = I.mat[q];
  return R;
}


/* what the world-box scans keep of a lane's item: whether it is a primitive with an exact pair routine (sphere 2, capsule 3,
 * box 6; feet, hull clouds and cylinders - two 16-gon rims, 32 vertices - are not), its centre in kernel coordinates and bounding
 * radius.  A handful of registers instead of the whole record: the item's frame and sizes are re-read from the model table
 * in the (rare) exact routine, so that nothing of the record stays live across the box loop. */
struct PrimLane { int ptype, code; float margin; bool cloud; /* lane = link geom: its contact with a world box comes from the vertex cloud scan */
                  const float* sph; /* LDS: the primitive's centre (kernel coordinates) and bounding radius - read per world box instead of riding in four registers */ };
/* [4 + GQ_MAXLG][4] behind the cloud spheres (GQ_BX_ISPH, gq_boxes.h) in the J block, idle until S7 */
#define GQ_BX_PSPH(W) (&(W).u.B[49][0])
__device__ __forceinline__ bool prim_exact(int ptype) { return ptype == 2 || ptype == 3 || ptype == 6; }
__device__ __forceinline__ PrimLane prim_lane(WaveMem& W, const GQ_MODEL GqDevModel& m, const ItemRegs& IT, const bool valid, const bool prims) {
  PrimLane P;
  const int lane = lane_id();
  P.cloud = true;
  if (prims) P.cloud = !prim_exact(m.lg[lane < m.nlg ? lane : 0].ptype);
  P.ptype = (valid && prim_exact(IT.ptype)) ? IT.ptype : 0; P.code = IT.code; P.margin = IT.margin;
  float* sph = GQ_BX_PSPH(W) + 4 * (lane < 4 + GQ_MAXLG ? lane : 0);
  P.sph = sph;
  if (P.ptype > 0) { /* only this lane reads its slot back: no barrier */
    st3(sph, ld3(W.xpos[IT.body]) + matvec(W.xmat[IT.body], IT.pos));
    sph[3] = IT.ptype == 6 ? sqrtf(IT.psize[0] * IT.psize[0] + IT.psize[1] * IT.psize[1] + IT.psize[2] * IT.psize[2]) : IT.psize[0] + (IT.ptype == 3 ? IT.psize[1] : 0.0f);
  }
  return P;
}

/* signed distance of point p (box frame) to the box of half sizes h, outward normal n (box frame); inside: nearest face */
__device__ __forceinline__ float point_box(V3 p, V3 h, V3& n) {
  const V3 q = v3(med3(p.x, -h.x, h.x), med3(p.y, -h.y, h.y), med3(p.z, -h.z, h.z));
  const V3 d = p - q;
  const float l2 = dot(d, d);
  if (l2 > 0.0f) { const float inv = fast_rsqrt(l2); n = inv * d; return l2 * inv; }
  const float ex = h.x - fabsf(p.x), ey = h.y - fabsf(p.y), ez = h.z - fabsf(p.z);
  if (ex <= ey && ex <= ez) { n = v3(p.x >= 0.0f ? 1.0f : -1.0f, 0.0f, 0.0f); return -ex; }
  if (ey <= ez) { n = v3(0.0f, p.y >= 0.0f ? 1.0f : -1.0f, 0.0f); return -ey; }
  n = v3(0.0f, 0.0f, p.z >= 0.0f ? 1.0f : -1.0f);
  return -ez;
}

/* capsule (axis p0-p1, radius r; a sphere when p0 == p1) against a box.  Point 1: the axis point closest to the box - the
 * derivative of the (convex, piecewise quadratic) squared distance along the axis is piecewise linear with kinks where a
 * coordinate crosses a face plane: evaluated at 0, 1 and the <= 6 kinks, the root lies between the last sample with a
 * non-positive and the first with a positive derivative.  An axis that passes through the box: the middle of the part
 * inside, pushed out through its nearest face.  Point 2: the end of the axis farther from point 1 if its own sphere is within the margin. */
__device__ inline void capsule_box(V3 p0, V3 p1, float r, V3 bc, const float* bR, V3 bh, float margin, PairHit& H) {
  H.n = 0;
  const V3 a = matTvec(bR, p0 - bc), b = matTvec(bR, p1 - bc), d = b - a;
  { /* the capsule's bounding sphere against the box: most candidates end here */
    V3 nn;
    if (point_box(a + 0.5f * d, bh, nn) - (r + 0.5f * sqrtf(dot(d, d))) >= margin) return;
  }
  auto gfun = [&](float sv, float& dep) {
    const V3 pp = a + sv * d;
    const V3 qq = v3(med3(pp.x, -bh.x, bh.x), med3(pp.y, -bh.y, bh.y), med3(pp.z, -bh.z, bh.z));
    dep = fminf(fminf(bh.x - fabsf(pp.x), bh.y - fabsf(pp.y)), bh.z - fabsf(pp.z));
    return dot(d, pp - qq);
  };
  float cand[6];
  bool cok[6];
  {
    const float ax[3] = {a.x, a.y, a.z}, dx[3] = {d.x, d.y, d.z}, hx[3] = {bh.x, bh.y, bh.z};
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
      for (int sg = 0; sg < 2; sg++) {
        const float sv = fabsf(dx[k]) > 1e-12f ? ((sg ? hx[k] : -hx[k]) - ax[k]) / dx[k] : -1.0f;
        cand[2 * k + sg] = sv; cok[2 * k + sg] = sv > 0.0f && sv < 1.0f;
      }
  }
  float dep0, dep1;
  const float g0 = gfun(0.0f, dep0), g1 = gfun(1.0f, dep1);
  const float epsg = 1e-5f * dot(d, d); /* an axis (numerically) parallel to the nearest face: the first end, whatever the round-off says */
  float lo = 0.0f, glo = g0, hi = 1.0f, ghi = g1;
#pragma unroll
  for (int i = 0; i < 6; i++) {
    float di;
    const float gi = gfun(cok[i] ? cand[i] : 0.0f, di);
    if (cok[i]) {
      if (gi < -epsg && cand[i] > lo) { lo = cand[i]; glo = gi; }   /* kinks where the derivative is (numerically) zero bound a flat stretch of */
      if (gi > epsg && cand[i] < hi) { hi = cand[i]; ghi = gi; }     /* equally close points: the secant across it picks one of them, reproducibly */
    }
  }
  float sstar = g0 >= -epsg ? 0.0f : (g1 <= epsg ? 1.0f : lo + (hi - lo) * (-glo) / (ghi - glo));
  { /* an axis that passes through the box (slab clipping): the middle of the part inside, pushed out through its nearest face */
    const float av[3] = {a.x, a.y, a.z}, dv[3] = {d.x, d.y, d.z}, hv[3] = {bh.x, bh.y, bh.z};
    float tE = 0.0f, tX = 1.0f;
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      if (fabsf(dv[k]) > 1e-12f) { const float s1 = (-hv[k] - av[k]) / dv[k], s2 = (hv[k] - av[k]) / dv[k]; tE = fmaxf(tE, fminf(s1, s2)); tX = fminf(tX, fmaxf(s1, s2)); }
      else if (fabsf(av[k]) > hv[k]) ok = false;
    }
    if (ok && tE < tX) sstar = 0.5f * (tE + tX);
  }
  (void)dep0; (void)dep1;
  /* (every PairHit slot is written with a compile-time index: a run-time one would move the record to scratch memory) */
  const bool sphere = d.x == 0.0f && d.y == 0.0f && d.z == 0.0f;
  {
    const V3 pp = a + sstar * d;
    V3 nl;
    const float dist = point_box(pp, bh, nl) - r;
    if (dist >= margin) return;
    const V3 nw = matvec(bR, nl);
    H.dist[0] = dist; H.nrm[0] = nw;
    H.pos[0] = bc + matvec(bR, pp) - (r + 0.5f * dist) * nw;
    H.n = 1;
  }
  const float s2 = sstar < 0.499f ? 1.0f : 0.0f; /* (a geom placed symmetrically has s* = 1/2 up to round-off: not a threshold to sit on) */
  if (sphere || fabsf(s2 - sstar) <= 1e-3f) return;
  {
    const V3 pp = a + s2 * d;
    V3 nl;
    const float dist = point_box(pp, bh, nl) - r;
    if (dist >= margin) return;
    const V3 nw = matvec(bR, nl);
    H.dist[1] = dist; H.nrm[1] = nw;
    H.pos[1] = bc + matvec(bR, pp) - (r + 0.5f * dist) * nw;
    H.n = 2;
  }
}

/* projection radius of a box (axes = columns of R, half sizes h) on the unit direction L */
__device__ __forceinline__ float box_radius(const float* R, V3 h, V3 L) {
  return h.x * fabsf(R[0] * L.x + R[3] * L.y + R[6] * L.z) + h.y * fabsf(R[1] * L.x + R[4] * L.y + R[7] * L.z) + h.z * fabsf(R[2] * L.x + R[5] * L.y + R[8] * L.z);
}
__device__ __forceinline__ V3 box_axis(const float* R, int i) { return v3(R[i], R[3 + i], R[6 + i]); }
__device__ __forceinline__ float sel3(int i, float x, float y, float z) { return i == 0 ? x : (i == 1 ? y : z); }

/* box A against box B: separating-axis test over the 15 axes; the axis of largest separation gives dist and the normal (an
 * edge-edge axis only if it beats the best face axis by more than 1e-6 + 5 %).  Face axis: the corners of the other box within
 * the margin of the reference face whose projection falls inside it (tolerance 1e-6) are the points, in corner order, the
 * deepest 4 kept; none: the corners of the reference face against the other box; no corner carries the axis depth: one more
 * point at the other box's support.  Edge axis: one point at the closest points of the two edges.  Normal from A to B. */
__device__ inline void box_box(V3 ca, const float* Ra, V3 ha, V3 cb, const float* Rb, V3 hb, float margin, PairHit& H) {
  H.n = 0;
  const V3 t = cb - ca;
  float best = -1e30f, beste = -1e30f;
  V3 bn = v3(0.0f, 0.0f, 1.0f), en = v3(0.0f, 0.0f, 1.0f);
  int bcode = 0, ei = -1, ej = 0;
#pragma unroll
  for (int w = 0; w < 2; w++)
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const V3 L = box_axis(w ? Rb : Ra, i);
      const float tl = dot(t, L), sep = fabsf(tl) - box_radius(Ra, ha, L) - box_radius(Rb, hb, L);
      if (sep > best + 2e-6f) { best = sep; bcode = 3 * w + i; bn = tl >= 0.0f ? L : -1.0f * L; } /* a later axis must win by more than fp32 noise */
    }
  if (best >= margin) return; /* a face axis separates the boxes: most candidate pairs end here, before the nine edge axes */
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      V3 L = cross(box_axis(Ra, i), box_axis(Rb, j));
      const float l2 = dot(L, L);
      if (l2 < 1e-2f) continue; /* (nearly) parallel edges: the closest points along them are ill-conditioned and the face axes describe the contact */
      L = fast_rsqrt(l2) * L;
      const float tl = dot(t, L), sep = fabsf(tl) - box_radius(Ra, ha, L) - box_radius(Rb, hb, L);
      if (sep > beste + 2e-6f) { beste = sep; ei = i; ej = j; en = tl >= 0.0f ? L : -1.0f * L; }
    }
  const bool edge = ei >= 0 && beste > best + 1e-6f + 0.05f * fabsf(best);
  const float sep = edge ? beste : best;
  if (sep >= margin) return;
  /* From here on every array index is a compile-time constant and box A / box B are told apart by value selects, never by
   * a pointer chosen at run time: either would move the matrices, the candidates and the caller's PairHit to scratch memory,
   * whose round trips then sit on the critical path of every wave (r03: 63.6 -> 51.5 M env-steps/s on aliengo). */
  if (edge) {
    V3 pa = ca, pb = cb;
    const float hav[3] = {ha.x, ha.y, ha.z}, hbv[3] = {hb.x, hb.y, hb.z};
    V3 ua = v3(0.0f, 0.0f, 0.0f), ub = ua;
    float hae = 0.0f, hbe = 0.0f;
#pragma unroll
    for (int q = 0; q < 3; q++) {
      const V3 aq = box_axis(Ra, q), bq = box_axis(Rb, q);
      const float da = dot(aq, en), db = dot(bq, en);
      if (q != ei) pa = pa + (fabsf(da) < 1e-4f ? 0.0f : (da >= 0.0f ? hav[q] : -hav[q])) * aq; else { ua = aq; hae = hav[q]; }
      if (q != ej) pb = pb + (fabsf(db) < 1e-4f ? 0.0f : (db >= 0.0f ? -hbv[q] : hbv[q])) * bq; else { ub = bq; hbe = hbv[q]; }
    }
    const V3 dp = pb - pa;
    const float uaub = dot(ua, ub), q1 = dot(ua, dp), q2 = -dot(ub, dp), den = 1.0f - uaub * uaub;
    float sa = den > 1e-12f ? (q1 + uaub * q2) / den : 0.0f, sb = den > 1e-12f ? (uaub * q1 + q2) / den : 0.0f;
    sa = med3(sa, -hae, hae); sb = med3(sb, -hbe, hbe);
    H.n = 1; H.dist[0] = sep; H.nrm[0] = en;
    H.pos[0] = 0.5f * ((pa + sa * ua) + (pb + sb * ub));
    return;
  }
  const bool refB = bcode >= 3;
  const int ax = bcode - (refB ? 3 : 0);
  const V3 cr = refB ? cb : ca, hr = refB ? hb : ha, ci = refB ? ca : cb, hi = refB ? ha : hb;
  float Rr[9], Ri[9];
#pragma unroll
  for (int q = 0; q < 9; q++) { Rr[q] = refB ? Rb[q] : Ra[q]; Ri[q] = refB ? Ra[q] : Rb[q]; }
  const V3 nr = refB ? -1.0f * bn : bn; /* outward normal of the reference face, pointing at the other box */
  const V3 rax = v3(sel3(ax, Rr[0], Rr[1], Rr[2]), sel3(ax, Rr[3], Rr[4], Rr[5]), sel3(ax, Rr[6], Rr[7], Rr[8]));
  const float sgn = dot(rax, nr) >= 0.0f ? 1.0f : -1.0f;
  const float hrax = sel3(ax, hr.x, hr.y, hr.z);
  /* candidate v of pass 0 (incident corners against the reference face) / pass 1 (reference-face corners against the other
   * box): distance, contact point, whether it counts.  Evaluated for all eight to rank them, again for the <= 4 kept. */
  auto corner = [&](const int v, const bool pass1, float& dd, V3& pt) -> bool {
    if (!pass1) {
      const V3 loc = v3((v & 1) ? hi.x : -hi.x, (v & 2) ? hi.y : -hi.y, (v & 4) ? hi.z : -hi.z);
      const V3 w = ci + matvec(Ri, loc);
      const V3 lr = matTvec(Rr, w - cr);
      dd = sgn * sel3(ax, lr.x, lr.y, lr.z) - hrax;
      const bool inside = (ax == 0 || fabsf(lr.x) <= hr.x + 1e-6f) && (ax == 1 || fabsf(lr.y) <= hr.y + 1e-6f) && (ax == 2 || fabsf(lr.z) <= hr.z + 1e-6f);
      pt = w - (0.5f * dd) * nr;
      return dd < margin && inside;
    }
    const float lax = ((v >> ax) & 1) ? 1.0f : -1.0f;
    const V3 loc = v3((v & 1) ? hr.x : -hr.x, (v & 2) ? hr.y : -hr.y, (v & 4) ? hr.z : -hr.z);
    const V3 w = cr + matvec(Rr, loc);
    V3 nl;
    dd = point_box(matTvec(Ri, w - ci), hi, nl);
    pt = w + (0.5f * dd) * nr;
    return lax * sgn >= 0.0f && dd < margin;
  };
  float cd[8];
#pragma unroll
  for (int k = 0; k < 8; k++) cd[k] = 1e30f;
  int mask = 0;
  bool pass1 = false;
#pragma unroll 1
  for (int pass = 0; pass < 2 && mask == 0; pass++) {
    pass1 = pass == 1;
#pragma unroll 1
    for (int v = 0; v < 8; v++) {
      float dd;
      V3 pt;
      const bool ok = corner(v, pass1, dd, pt);
#pragma unroll
      for (int k = 0; k < 8; k++) cd[k] = v == k ? dd : cd[k];
      if (ok) mask |= 1 << v;
    }
  }
  float mincd = 1e30f;
#pragma unroll
  for (int v = 0; v < 8; v++) mincd = ((mask >> v) & 1) ? fminf(mincd, cd[v]) : mincd;
  int ncand = __builtin_popcount(mask);
  const bool extra = ncand == 0 || mincd > sep + 1e-4f;
  V3 xp = ci;
  {
    const float hiv[3] = {hi.x, hi.y, hi.z};
#pragma unroll
    for (int q = 0; q < 3; q++) { /* support point; an axis (numerically) parallel to the face contributes its midpoint */
      const V3 iq = box_axis(Ri, q);
      const float dq = dot(iq, nr);
      xp = xp + (fabsf(dq) < 1e-4f ? 0.0f : (dq >= 0.0f ? -hiv[q] : hiv[q])) * iq;
    }
    xp = xp - (0.5f * sep) * nr;
  }
  if (extra && ncand == 8) { /* (cannot happen: eight corners inside the margin carry the depth) drop the last */
    mask &= 0x7f; ncand = 7;
  }
  /* the deepest 4 in candidate order (the extra point comes last): drop the shallowest while more than 4 are left */
  int total = ncand + (extra ? 1 : 0);
  bool keepx = extra;
#pragma unroll 1
  for (; total > 4; total--) {
    float wv = -1e30f;
    int worst = 0;
#pragma unroll
    for (int v = 0; v < 8; v++) if (((mask >> v) & 1) && cd[v] >= wv) { wv = cd[v]; worst = v; }
    if (keepx && sep >= wv) { keepx = false; }
    else mask &= ~(1 << worst);
  }
#pragma unroll
  for (int k = 0; k < 4; k++) { H.dist[k] = 0.0f; H.pos[k] = v3(0.0f, 0.0f, 0.0f); }
  H.nrm[0] = bn; H.nrm[1] = bn;
  int n = 0;
#pragma unroll 1
  for (; mask != 0; n++) {
    const int v = __builtin_ctz(mask);
    mask &= mask - 1;
    float dd;
    V3 pt;
    corner(v, pass1, dd, pt);
#pragma unroll
    for (int k = 0; k < 4; k++) { H.dist[k] = n == k ? dd : H.dist[k]; H.pos[k].x = n == k ? pt.x : H.pos[k].x; H.pos[k].y = n == k ? pt.y : H.pos[k].y; H.pos[k].z = n == k ? pt.z : H.pos[k].z; }
  }
  if (keepx) {
#pragma unroll
    for (int k = 0; k < 4; k++) { H.dist[k] = n == k ? sep : H.dist[k]; H.pos[k].x = n == k ? xp.x : H.pos[k].x; H.pos[k].y = n == k ? xp.y : H.pos[k].y; H.pos[k].z = n == k ? xp.z : H.pos[k].z; }
    n++;
  }
  H.n = n;
}

}  // namespace gq
