/*
 * gq_convex.h - the general convex narrow phase of the step kernel: what MuJoCo's mjc_Convex computes for a pair of convex geoms
 * (mj_collision inside mj_step, quadruped_env.py:271) - signed distance / penetration depth, the normal of the minimum translation
 * and the point midway between the two witness points, ONE contact per pair (multiccd is off by default) - for the robot's mesh hulls
 * and cylinders against the world boxes of a scene and against each other.  GJK for the distance of the un-inflated cores, the
 * expanding polytope algorithm when they overlap; restated step for step in oracle/gq_convex.h (fp64), which is pinned against the
 * exact Minkowski-difference hull (tests/test_oracle_invariants.py).
 *
 * One wavefront works on one pair:
 *   support query   lane = vertex: the masked 64-vertex chunks of the hull's DIRECTION-ordered copy (the cube-map table of the plane
 *                   narrow phase, GqDevGeom::pmask_adr: ~2 of up to 11 chunks hold the support vertex of a direction), DPP wave-max, the
 *                   winner's coordinates by v_readlane; world boxes and sphere / capsule cores answer analytically
 *   GJK simplex     wave-uniform arithmetic (<= 4 points kept in LDS)
 *   EPA polytope    lane = face (<= 4 + 2 x 24 faces): closest face by DPP wave-min; the faces that see the new vertex grow as a
 *                   connected patch by ballots over the faces' neighbour words; rim edges ranked by one wave prefix scan; the fan's
 *                   faces take the freed lanes
 * Scratch: 200 words of the idle J block (polytope vertices, face neighbours, rim list) and 48 of the idle factor block (the two
 * shape descriptors, the result) - the caller says where, see gq_boxes.h.
 */
#pragma once
#include "gq_step_kernel.h"

namespace gq {

/* The routine is INLINED into the kernel: as a function of its own (one copy of the code, two call sites) its register need is not bounded by
 * the kernel's __launch_bounds__ - the step kernels that call it were allocated 165 VGPRs, two waves per SIMD instead of four, and a 4 096-env
 * batch was no longer co-resident (launch time x 3).  -DGQ_CVX_CALL keeps the call for experiments. */
#ifdef GQ_CVX_CALL
#define GQ_CVX_FN __device__ __attribute__((noinline))
#else
#define GQ_CVX_FN __device__ __forceinline__
#endif
#ifdef GQ_CVX_STATS /* tools/ubench/convex_pair.hip: cycle sums per part of the routine */
extern __device__ long long gq_cvx_cyc[8];
#define GQ_CVX_T(i) do { const long long t_ = __builtin_readcyclecounter(); if (lane_id() == 0) gq_cvx_cyc[i] += t_ - tk_; tk_ = t_; } while (0)
#define GQ_CVX_T0() long long tk_ = __builtin_readcyclecounter()
#else
#define GQ_CVX_T(i) do { } while (0)
#define GQ_CVX_T0() do { } while (0)
#endif
#ifndef GQ_CVX_BATCH
#define GQ_CVX_BATCH 2  /* chunks per shape whose vertex loads go out together (three: 18 registers of loads in flight, spilled by the kernels that inline the routine) */
#endif
#define GQ_CVX_GJK_MAXIT 32
#define GQ_CVX_EPA_MAXIT 24
#define GQ_CVX_MAXV 28                 /* polytope vertices: 4 + one per EPA iteration */
#define GQ_CVX_MAXRIM 24
#define GQ_CVX_POLY_WORDS (4 * GQ_CVX_MAXV + 64 + GQ_CVX_MAXRIM)  /* 200 */
#define GQ_CVX_SHAPE_WORDS 20
#define GQ_CVX_SHP_WORDS (2 * GQ_CVX_SHAPE_WORDS + 8)
#define GQ_CVX_TOL_GJK 1e-6f           /* relative, on v.v - v.w */
#define GQ_CVX_TOL_EPA 1e-8f           /* metres: a face whose support point lies no farther out is a face of A - B (the round-off of fp32 coordinates about the base is 6e-8; the usual exit is exact - the support vertex is already a vertex of the polytope) */

/* a shape as the routine sees it (wave-uniform, in LDS): kind 0 vertex cloud [adr, adr + num) of the vertex arrays in the frame (R, t),
 * pm = index of its chunk caps in the vertex arrays (GqDevGeom::cap_adr) or -1; kind 1 box, centre t, axes = columns of R, half extents h; kind 2 segment t .. h
 * (capsule / sphere core, world end points).  r: the radius that inflates the core. */
struct CvxShape { int kind, adr, num, pm; float R[9]; V3 t, h; float r; };
/* the routine is a function of its own (one copy per kernel, two call sites): its scratch pointers carry the LDS address space in their
 * type, or every access through them would be a FLAT instruction */
typedef GQ_LDS float* LdsF;
typedef const GQ_LDS float* LdsCF;
typedef GQ_LDS int32_t* LdsI;
typedef const GQ_LDS int32_t* LdsCI;
__device__ __forceinline__ void st3l(LdsF p, V3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
__device__ __forceinline__ void cvx_shape_store(LdsF S, const CvxShape& s) { /* (every lane stores the same words) */
  LdsI I = (LdsI)S;
  I[0] = s.kind; I[1] = s.adr; I[2] = s.num; I[3] = s.pm;
#pragma unroll
  for (int i = 0; i < 9; i++) S[4 + i] = s.R[i];
  st3l(S + 13, s.t); st3l(S + 16, s.h); S[19] = s.r;
}

/* per-lane copy of the two shapes' chunk caps (lane k < 16: chunk k of A, lane 16 + k: chunk k of B): the support vertex of direction d
 * can only lie in a chunk whose cap - axis, cosine of the half angle; host: cabi.plane_support_tables, the enclosing cap of the normal
 * cones of the chunk's vertices in the DIRECTION-ordered copy - contains d.  One load per pair, at the start of the routine: a support
 * query then costs ONE memory round trip (the vertex loads of both shapes' chunks, issued together) instead of a table word and then the
 * vertices, per shape one after the other - the pair routine is a chain of such queries and nothing else */
struct CvxCaps { V3 ax; float cs; };
__device__ inline CvxCaps cvx_caps_fetch(LdsCF SA, LdsCF SB, const GQ_MODEL float* vx, const GQ_MODEL float* vy, const GQ_MODEL float* vz) {
  const int lane = lane_id();
  LdsCI IA = (LdsCI)SA; LdsCI IB = (LdsCI)SB;
  const int k = lane & 15;
  const int capA = uniform(IA[3]), capB = uniform(IB[3]);
  const int c = lane < 16 ? capA : capB;
  CvxCaps o;
  o.ax = v3(0.0f, 0.0f, 0.0f); o.cs = -2.0f; /* no caps: every chunk passes */
  if (lane < 32 && c >= 0) { o.ax = v3(vx[c + k], vy[c + k], vz[c + k]); o.cs = vx[c + 16 + k]; }
  return o;
}

/* support point of A - B in world direction d: w = s_A(d) - s_B(-d) and the packed vertex ids (ia | ib << 16).  Clouds: lane = vertex
 * over the chunks their caps let through, the loads of BOTH shapes in flight together, DPP wave-max, the winners' coordinates by v_readlane;
 * boxes and segments answer analytically.  (One copy of this code per kernel.) */
struct CvxMink { V3 w; int id; };
/* the two shapes' descriptors in registers for the length of the routine (wave-uniform values): a query read them from LDS - 40 words, their
 * latency in front of every one of the ten to twenty queries of a pair */
struct CvxRegs { int kA, kB, adrA, adrB, lastA, lastB, chA, chB; LdsCF RA, RB, tA_, tB_, hA_, hB_; };
__device__ __forceinline__ CvxRegs cvx_regs(LdsCF SA, LdsCF SB) {
  LdsCI IA = (LdsCI)SA; LdsCI IB = (LdsCI)SB;
  CvxRegs G;
  G.kA = uniform(IA[0]); G.kB = uniform(IB[0]);
  G.adrA = uniform(IA[1]); G.adrB = uniform(IB[1]);
  const int nA = uniform(IA[2]), nB = uniform(IB[2]);
  G.lastA = G.adrA + nA - 1; G.lastB = G.adrB + nB - 1;
  G.chA = (1 << ((nA + GQ_WAVE - 1) / GQ_WAVE)) - 1; G.chB = (1 << ((nB + GQ_WAVE - 1) / GQ_WAVE)) - 1;
  G.RA = SA + 4; G.RB = SB + 4; G.tA_ = SA + 13; G.tB_ = SB + 13; G.hA_ = SA + 16; G.hB_ = SB + 16;
  return G;
}
GQ_CVX_FN CvxMink cvx_minkowski(const CvxRegs& G, const GQ_MODEL float* vx, const GQ_MODEL float* vy, const GQ_MODEL float* vz, const V3 d, const CvxCaps caps) {
  const int lane = lane_id();
  const int kA = G.kA, kB = G.kB;
  const V3 dlA = matTvec(G.RA, d), dlB = matTvec(G.RB, -1.0f * d);
  int cmA = 0, cmB = 0;
  const int adrA = G.adrA, adrB = G.adrB, lastA = G.lastA, lastB = G.lastB;
  if (kA == 0 || kB == 0) { /* wave-uniform */
    const V3 mine = lane < 16 ? dlA : dlB;
    const float inv = fast_rsqrt(fmaxf(dot(mine, mine), 1e-30f));
    const uint64_t pass = ballot(inv * dot(mine, caps.ax) >= caps.cs - 1e-4f);
    if (kA == 0) cmA = (int)(pass & 0xffffull) & G.chA;
    if (kB == 0) cmB = (int)((pass >> 16) & 0xffffull) & G.chB;
  }
  float bestA = -3e38f, bestB = -3e38f;
  V3 pA = v3(0.0f, 0.0f, 0.0f), pB = v3(0.0f, 0.0f, 0.0f);
  int iA = adrA, iB = adrB;
  while (cmA | cmB) { /* wave-uniform: up to GQ_CVX_BATCH chunks of each shape per trip, their loads in flight together; lanes past the end re-read the last vertex */
    int cu[2 * GQ_CVX_BATCH];
    float px[2 * GQ_CVX_BATCH], py[2 * GQ_CVX_BATCH], pz[2 * GQ_CVX_BATCH];
#pragma unroll
    for (int u = 0; u < GQ_CVX_BATCH; u++) { cu[u] = cmA ? __builtin_ctz(cmA) : -1; cmA &= cmA - 1; }
#pragma unroll
    for (int u = GQ_CVX_BATCH; u < 2 * GQ_CVX_BATCH; u++) { cu[u] = cmB ? __builtin_ctz(cmB) : -1; cmB &= cmB - 1; }
#pragma unroll
    for (int u = 0; u < 2 * GQ_CVX_BATCH; u++)
      if (cu[u] >= 0) {
        const int i = (u < GQ_CVX_BATCH ? adrA : adrB) + cu[u] * GQ_WAVE + lane, last = u < GQ_CVX_BATCH ? lastA : lastB, ii = i < last ? i : last;
        px[u] = vx[ii]; py[u] = vy[ii]; pz[u] = vz[ii];
      }
#pragma unroll
    for (int u = 0; u < GQ_CVX_BATCH; u++)
      if (cu[u] >= 0) {
        const int i = adrA + cu[u] * GQ_WAVE + lane, ii = i < lastA ? i : lastA;
        const float pr = dlA.x * px[u] + dlA.y * py[u] + dlA.z * pz[u];
        if (pr > bestA) { bestA = pr; pA = v3(px[u], py[u], pz[u]); iA = ii; }
      }
#pragma unroll
    for (int u = GQ_CVX_BATCH; u < 2 * GQ_CVX_BATCH; u++)
      if (cu[u] >= 0) {
        const int i = adrB + cu[u] * GQ_WAVE + lane, ii = i < lastB ? i : lastB;
        const float pr = dlB.x * px[u] + dlB.y * py[u] + dlB.z * pz[u];
        if (pr > bestB) { bestB = pr; pB = v3(px[u], py[u], pz[u]); iB = ii; }
      }
  }
  V3 a, b;
  int ida, idb;
  if (kA == 0) {
    const float wmax = wave_max(bestA);
    const int who = ffs64(ballot(bestA == wmax));
    a = ld3(G.tA_) + matvec(G.RA, v3(bcast(pA.x, who), bcast(pA.y, who), bcast(pA.z, who)));
    ida = bcast(iA, who) - adrA;
  } else if (kA == 1) {
    const V3 h = ld3(G.hA_);
    ida = (dlA.x < 0.0f ? 1 : 0) | (dlA.y < 0.0f ? 2 : 0) | (dlA.z < 0.0f ? 4 : 0);
    a = ld3(G.tA_) + matvec(G.RA, v3(dlA.x < 0.0f ? -h.x : h.x, dlA.y < 0.0f ? -h.y : h.y, dlA.z < 0.0f ? -h.z : h.z));
  } else {
    const V3 p0 = ld3(G.tA_), p1 = ld3(G.hA_);
    const bool far = dot(d, p1 - p0) > 0.0f;
    a = far ? p1 : p0; ida = far ? 1 : 0;
  }
  if (kB == 0) {
    const float wmax = wave_max(bestB);
    const int who = ffs64(ballot(bestB == wmax));
    b = ld3(G.tB_) + matvec(G.RB, v3(bcast(pB.x, who), bcast(pB.y, who), bcast(pB.z, who)));
    idb = bcast(iB, who) - adrB;
  } else if (kB == 1) {
    const V3 h = ld3(G.hB_);
    idb = (dlB.x < 0.0f ? 1 : 0) | (dlB.y < 0.0f ? 2 : 0) | (dlB.z < 0.0f ? 4 : 0);
    b = ld3(G.tB_) + matvec(G.RB, v3(dlB.x < 0.0f ? -h.x : h.x, dlB.y < 0.0f ? -h.y : h.y, dlB.z < 0.0f ? -h.z : h.z));
  } else {
    const V3 p0 = ld3(G.tB_), p1 = ld3(G.hB_);
    const bool far = dot(d, p1 - p0) < 0.0f;
    b = far ? p1 : p0; idb = far ? 1 : 0;
  }
  CvxMink o;
  o.w = a - b; o.id = ida | (idb << 16);
  return o;
}
/* the point of shape S that cvx_support returned with `id` */
__device__ inline V3 cvx_point(LdsCF S, const GQ_MODEL float* vx, const GQ_MODEL float* vy, const GQ_MODEL float* vz, const int id) {
  LdsCI I = (LdsCI)S;
  const int kind = uniform(I[0]);
  if (kind == 2) return id ? ld3(S + 16) : ld3(S + 13);
  LdsCF R = S + 4;
  const V3 t = ld3(S + 13);
  if (kind == 1) {
    const V3 h = ld3(S + 16);
    return t + matvec(R, v3((id & 1) ? -h.x : h.x, (id & 2) ? -h.y : h.y, (id & 4) ? -h.z : h.z));
  }
  const int i = uniform(I[1]) + id;
  return t + matvec(R, v3(vx[i], vy[i], vz[i]));
}

/* closest point of a segment / triangle to the origin as barycentric weights (Ericson 5.1.2 / 5.1.5; oracle cvx_seg / cvx_tri).
 * In DOUBLE precision on fp32 points: a simplex of A - B is routinely a sliver - two vertices of a mesh a few millimetres apart
 * against a corner of a world box a metre away - and the weights of the nearest point are ratios of differences of products of its
 * edge vectors; in fp32 the point came out 1.3e-4 m off on such a sliver (a tenth of the contact margin) and GJK stalled there.
 * The arithmetic is wave-uniform and a few dozen operations per iteration; gfx950 issues v_fma_f64 at the fp32 rate. */
struct D3 { double x, y, z; };
__device__ __forceinline__ D3 d3(V3 a) { D3 r = {(double)a.x, (double)a.y, (double)a.z}; return r; }
__device__ __forceinline__ D3 operator-(D3 a, D3 b) { D3 r = {a.x - b.x, a.y - b.y, a.z - b.z}; return r; }
__device__ __forceinline__ double dot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ D3 cross(D3 a, D3 b) { D3 r = {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; return r; }
__device__ __forceinline__ double bcastd(double v, int src) { /* v_readlane of both halves */
  struct I2 { int x, y; };
  I2 u = __builtin_bit_cast(I2, v);
  u.x = bcast(u.x, src); u.y = bcast(u.y, src);
  return __builtin_bit_cast(double, u);
}
__device__ __forceinline__ void cvx_seg(D3 p0, D3 p1, double* lam) {
  const D3 e = p1 - p0;
  const double ee = dot(e, e), t = ee > 0.0 ? -dot(p0, e) / ee : 0.0;
  const double tc = t <= 0.0 ? 0.0 : (t >= 1.0 ? 1.0 : t);
  lam[0] = 1.0 - tc; lam[1] = tc;
}
__device__ inline void cvx_tri(D3 a, D3 b, D3 c, double* lam) {
  const D3 ab = b - a, ac = c - a;
  const double d1 = -dot(ab, a), d2 = -dot(ac, a);
  lam[0] = lam[1] = lam[2] = 0.0;
  if (d1 <= 0.0 && d2 <= 0.0) { lam[0] = 1.0; return; }
  const double d3_ = -dot(ab, b), d4 = -dot(ac, b);
  if (d3_ >= 0.0 && d4 <= d3_) { lam[1] = 1.0; return; }
  const double vc = d1 * d4 - d3_ * d2;
  if (vc <= 0.0 && d1 >= 0.0 && d3_ <= 0.0) { const double v = d1 / (d1 - d3_); lam[0] = 1.0 - v; lam[1] = v; return; }
  const double d5 = -dot(ab, c), d6 = -dot(ac, c);
  if (d6 >= 0.0 && d5 <= d6) { lam[2] = 1.0; return; }
  const double vb = d5 * d2 - d1 * d6;
  if (vb <= 0.0 && d2 >= 0.0 && d6 <= 0.0) { const double w = d2 / (d2 - d6); lam[0] = 1.0 - w; lam[2] = w; return; }
  const double va = d3_ * d6 - d5 * d4;
  if (va <= 0.0 && (d4 - d3_) >= 0.0 && (d5 - d6) >= 0.0) { const double w = (d4 - d3_) / ((d4 - d3_) + (d5 - d6)); lam[1] = 1.0 - w; lam[2] = w; return; }
  const double den = 1.0 / (va + vb + vc);
  lam[1] = vb * den; lam[2] = vc * den; lam[0] = 1.0 - lam[1] - lam[2];
}
__device__ __forceinline__ V3 cvx_comb(const double* l, D3 a, D3 b, D3 c) {
  return v3((float)(l[0] * a.x + l[1] * b.x + l[2] * c.x), (float)(l[0] * a.y + l[1] * b.y + l[2] * c.y), (float)(l[0] * a.z + l[1] * b.z + l[2] * c.z));
}

/* polytope vertex k in the scratch: w (3 words), ids (ia | ib << 16) */
#define GQ_CVX_PW(P, k) ld3((P) + 4 * (k))
#define GQ_CVX_PID(P, k) (((LdsCI)(P))[4 * (k) + 3])

/* closest point of the simplex P[0..n) to the origin: weights lam (zero where a vertex is not needed), the point v; true when a
 * tetrahedron encloses the origin (oracle cvx_simplex) */
__device__ inline bool cvx_simplex(LdsCF P, const int n, float* lam, V3& v) {
  lam[0] = lam[1] = lam[2] = lam[3] = 0.0f;
  const V3 f0 = GQ_CVX_PW(P, 0);
  if (n == 1) { lam[0] = 1.0f; v = f0; return false; }
  const D3 p0 = d3(f0), p1 = d3(GQ_CVX_PW(P, 1));
  if (n == 2) { double l2[3]; cvx_seg(p0, p1, l2); l2[2] = 0.0; lam[0] = (float)l2[0]; lam[1] = (float)l2[1]; v = cvx_comb(l2, p0, p1, p1); return false; }
  const D3 p2 = d3(GQ_CVX_PW(P, 2));
  if (n == 3) { double l3[3]; cvx_tri(p0, p1, p2, l3); lam[0] = (float)l3[0]; lam[1] = (float)l3[1]; lam[2] = (float)l3[2]; v = cvx_comb(l3, p0, p1, p2); return false; }
  const D3 p3 = d3(GQ_CVX_PW(P, 3));
  /* the tetrahedron, lane f & 3 = face f: (0,1,2 | 3), (0,1,3 | 2), (0,2,3 | 1), (1,2,3 | 0) - the four faces side by side instead of one
   * after the other on every lane (the fp64 triangle routine is ~100 dependent instructions, and this is the iteration's critical path);
   * the nearest of the faces the origin lies beyond wins, the first of equals in face order like the oracle's loop */
  const int f = lane_id() & 3;
  const D3 a = f == 3 ? p1 : p0, b = f < 2 ? p1 : p2, c = f == 0 ? p2 : p3, o = f == 0 ? p3 : (f == 1 ? p2 : (f == 2 ? p1 : p0));
  const D3 nf = cross(b - a, c - a);
  const double so = dot(nf, o - a), sz = -dot(nf, a);
  const bool beyond = (so > 0.0 && sz < 0.0) || (so < 0.0 && sz > 0.0) || so == 0.0;
  double l3[3];
  cvx_tri(a, b, c, l3);
  const V3 p = cvx_comb(l3, a, b, c);
  const double pp = beyond ? (double)p.x * p.x + (double)p.y * p.y + (double)p.z * p.z : 1e300;
  double best = 1e300;
  int fb = -1;
#pragma unroll
  for (int k = 0; k < 4; k++) { const double ppk = bcastd(pp, k); if (ppk < best) { best = ppk; fb = k; } }
  if (fb < 0) return true; /* the origin is beyond no face: enclosed */
  v = v3(bcast(p.x, fb), bcast(p.y, fb), bcast(p.z, fb));
  const float w0 = (float)bcastd(l3[0], fb), w1 = (float)bcastd(l3[1], fb), w2 = (float)bcastd(l3[2], fb);
  lam[0] = fb == 3 ? 0.0f : w0;
  lam[1] = fb < 2 ? w1 : (fb == 3 ? w0 : 0.0f);
  lam[2] = fb == 0 ? w2 : (fb == 1 ? 0.0f : w1);
  lam[3] = fb == 0 ? 0.0f : w2;
  return false;
}

/* unit normal and offset of the polytope face (a, b, c); a sliver (edges parallel to 1e-5) gets the offset 1e30: kept for the topology,
 * never the closest face (oracle cvx_face_plane) */
__device__ __forceinline__ void cvx_face_plane(V3 a, V3 b, V3 c, V3& n, float& d) {
  /* (double precision: the vertices of A - B lie up to a metre from the origin - a corner of a world box - while the offset wanted is
   * millimetres to 1e-6; an fp32 cross product of a thin face tilts its normal by 1e-3 and moves the offset by 1e-5) */
  const D3 A = d3(a), ab = d3(b) - A, ac = d3(c) - A, x = cross(ab, ac);
  const double l2 = dot(x, x);
  if (l2 > 1e-10 * dot(ab, ab) * dot(ac, ac) && l2 > 1e-60) {
    double inv;
    if (l2 > 1e-30) { const double r = (double)fast_rsqrt((float)l2); inv = r * (1.5 - 0.5 * l2 * r * r); } /* v_rsq_f32 + one Newton step in fp64 (1e-14): fp64 sqrt and division are dozens of instructions each */
    else inv = 1.0 / sqrt(l2);
    n = v3((float)(x.x * inv), (float)(x.y * inv), (float)(x.z * inv));
    d = (float)(dot(x, A) * inv);
  } else { n = v3(0.0f, 0.0f, 1.0f); d = 1e30f; }
}

/* One convex pair.  shp: the two shape descriptors (A, B: GQ_CVX_SHAPE_WORDS each), then the result - dist, normal A -> B (3), point (3);
 * poly: GQ_CVX_POLY_WORDS words of scratch.  Returns true when the inflated shapes are closer than margin; false: result words 1..3 hold the
 * separating direction it found (0 0 0 when it has none to offer).  Wave-uniform. */
GQ_CVX_FN bool cvx_pair_wave(LdsF shp, LdsF poly, const GQ_MODEL float* vx, const GQ_MODEL float* vy, const GQ_MODEL float* vz, const float margin, const V3 hint = {0.0f, 0.0f, 0.0f}) {
  const int lane = lane_id();
  LdsCF SA = shp; LdsCF SB = shp + GQ_CVX_SHAPE_WORDS;
  LdsF out = shp + 2 * GQ_CVX_SHAPE_WORDS;
  LdsF P = poly;
  LdsI PI = (LdsI)poly;
  LdsI ADJ = PI + 4 * GQ_CVX_MAXV;
  LdsI RIM = ADJ + 64;
  const float rA = SA[19], rB = SB[19], reach = margin + rA + rB;
  st3l(out + 1, v3(0.0f, 0.0f, 0.0f)); /* a pair found APART leaves the direction (A -> B) that separates it by more than margin + radii here: the caller's axis cache */
  const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  GQ_CVX_T0();
  const CvxCaps caps = cvx_caps_fetch(SA, SB, vx, vy, vz);
  const CvxRegs G = cvx_regs(SA, SB);
  /* ---- GJK */
  V3 v;
  float lam[4] = {1.0f, 0.0f, 0.0f, 0.0f};
  int ns = 1;
  {
    V3 d0 = (uniform(((LdsCI)SB)[0]) == 2 ? 0.5f * (ld3(SB + 13) + ld3(SB + 16)) : ld3(SB + 13)) -
            (uniform(((LdsCI)SA)[0]) == 2 ? 0.5f * (ld3(SA + 13) + ld3(SA + 16)) : ld3(SA + 13));
    if (dot(hint, hint) > 0.0f) d0 = hint; /* the caller's first search direction, from A to B (world boxes: out of the box towards the cloud's centre) */
    if (dot(d0, d0) < 1e-24f) d0 = v3(1.0f, 0.0f, 0.0f);
    const CvxMink s0 = cvx_minkowski(G, vx, vy, vz, d0, caps);
    v = s0.w;
    if (-dot(v, d0) > reach * fast_sqrt(dot(d0, d0))) { st3l(out + 1, d0); return false; } /* the first direction already separates the cores by more than reach */
    wave_barrier();
    st3l(P, v); PI[3] = s0.id;
    wave_barrier();
  }
  bool enclosed = false;
#pragma unroll 1
  for (int it = 0; it < GQ_CVX_GJK_MAXIT; it++) {
    const float vv = dot(v, v);
    if (vv < 1e-24f) { enclosed = true; break; } /* the origin lies on the simplex: touching cores */
    GQ_CVX_T(0);
    const CvxMink sw = cvx_minkowski(G, vx, vy, vz, -1.0f * v, caps);
    GQ_CVX_T(1);
    const V3 w = sw.w;
    const float vw = dot(v, w);
    if (vw > 0.0f && vw * vw > reach * reach * vv) { st3l(out + 1, -1.0f * v); return false; } /* the cores are farther apart than anything of interest */
    const int wid = sw.id;
    bool dup = false;
#pragma unroll
    for (int i = 0; i < 4; i++) dup = dup || (i < ns && GQ_CVX_PID(P, i) == wid);
#ifdef GQ_EMU_TRACE
    if (lane == 0 && getenv("GQ_CVX_TRACE")) printf("gjk it %d ns %d vv %.9g vw %.9g dup %d wid %x\n", it, ns, (double)vv, (double)vw, (int)dup, wid);
#endif
    if (dup || vv - vw <= GQ_CVX_TOL_GJK * (vv + fast_sqrt(vv * dot(w, w)))) break; /* v is the closest point */
    wave_barrier();
    st3l(P + 4 * ns, w); PI[4 * ns + 3] = wid;
    ns++;
    wave_barrier();
#ifdef GQ_EMU_TRACE
    if (lane == 0 && getenv("GQ_CVX_TRACE") && it >= 6 && it <= 8) for (int i = 0; i < ns; i++) printf("   S[%d] = %.9g %.9g %.9g id %x\n", i, (double)P[4 * i], (double)P[4 * i + 1], (double)P[4 * i + 2], GQ_CVX_PID(P, i));
#endif
    if (cvx_simplex(P, ns, lam, v)) { enclosed = true; break; }
#ifdef GQ_EMU_TRACE
    if (lane == 0 && getenv("GQ_CVX_TRACE") && it >= 6 && it <= 8) printf("   lam %.9g %.9g %.9g %.9g v %.9g %.9g %.9g\n", (double)lam[0], (double)lam[1], (double)lam[2], (double)lam[3], (double)v.x, (double)v.y, (double)v.z);
#endif
    /* keep the vertices that carry the point (wave-uniform data: every lane moves the same words) */
    V3 kw[4]; int kid[4]; float kl[4]; int m = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) { kw[i] = v3(0.0f, 0.0f, 0.0f); kid[i] = 0; kl[i] = 0.0f; }
#pragma unroll
    for (int i = 0; i < 4; i++)
      if (i < ns && lam[i] > 0.0f) {
        const V3 pw = GQ_CVX_PW(P, i); const int pid = GQ_CVX_PID(P, i);
#pragma unroll
        for (int q = 0; q < 4; q++) if (q == m) { kw[q] = pw; kid[q] = pid; kl[q] = lam[i]; }
        m++;
      }
    wave_barrier();
#pragma unroll
    for (int q = 0; q < 4; q++) if (q < m) { st3l(P + 4 * q, kw[q]); PI[4 * q + 3] = kid[q]; lam[q] = kl[q]; } else lam[q] = 0.0f;
    ns = m;
    wave_barrier();
    GQ_CVX_T(2);
  }
  GQ_CVX_T(0);
  if (!enclosed) {
    const float len = fast_sqrt(dot(v, v));
    const float dist = len - rA - rB;
    if (!(dist < margin)) { st3l(out + 1, -1.0f * v); return false; }
    const V3 n = fast_rcp(len) * (-1.0f * v);
    V3 pa = v3(0.0f, 0.0f, 0.0f);
#pragma unroll
    for (int i = 0; i < 4; i++)
      if (i < ns && lam[i] > 0.0f) pa = pa + lam[i] * cvx_point(SA, vx, vy, vz, GQ_CVX_PID(P, i) & 0xffff);
    const V3 pb = pa - v;
    const V3 pos = 0.5f * ((pa + rA * n) + (pb - rB * n));
    wave_barrier();
    out[0] = dist; st3l(out + 1, n); st3l(out + 4, pos);
    wave_barrier();
    return true;
  }
#ifdef GQ_CVX_STATS
  ((LdsI)out)[7] = 0;
#endif
  /* ---- the cores overlap.  A tetrahedron around the origin first: a touching / degenerate simplex is blown up with supports along
   * directions it does not span (oracle: same order of attempts) */
  if (ns < 4) {
#pragma unroll 1
    for (int tries = 0; tries < 12 && ns < 4; tries++) {
      V3 dir;
      const V3 p0 = GQ_CVX_PW(P, 0);
      const int k = tries % 3;
      const V3 axk = v3(k == 0 ? 1.0f : 0.0f, k == 1 ? 1.0f : 0.0f, k == 2 ? 1.0f : 0.0f);
      if (ns == 1) dir = tries < 3 ? axk : -1.0f * axk;
      else if (ns == 2) { dir = cross(GQ_CVX_PW(P, 1) - p0, axk); if (tries >= 3 && tries < 6) dir = -1.0f * dir; }
      else { dir = cross(GQ_CVX_PW(P, 1) - p0, GQ_CVX_PW(P, 2) - p0); if (tries & 1) dir = -1.0f * dir; }
      if (dot(dir, dir) < 1e-30f) continue;
      const CvxMink sw = cvx_minkowski(G, vx, vy, vz, dir, caps);
      const V3 w = sw.w;
      const int wid = sw.id;
      bool dup = false;
#pragma unroll
      for (int i = 0; i < 4; i++) dup = dup || (i < ns && GQ_CVX_PID(P, i) == wid);
      if (dup) continue;
      const V3 f = w - p0;
      if (ns == 1) { if (dot(f, f) < 1e-20f) continue; }
      else if (ns == 2) { const V3 e = GQ_CVX_PW(P, 1) - p0, c = cross(e, f); if (dot(c, c) < 1e-12f * dot(e, e) * dot(f, f)) continue; }
      else { const V3 c = cross(GQ_CVX_PW(P, 1) - p0, GQ_CVX_PW(P, 2) - p0); const float vol = dot(c, f); if (vol * vol < 1e-12f * dot(c, c) * dot(f, f)) continue; }
      wave_barrier();
      st3l(P + 4 * ns, w); PI[4 * ns + 3] = wid;
      ns++;
      wave_barrier();
    }
    if (ns < 4) return false; /* a flat Minkowski difference: no volume to penetrate */
  }
  /* ---- EPA, lane = face */
  int fv = 0;                    /* v0 | v1 << 8 | v2 << 16 */
  V3 fn = v3(0.0f, 0.0f, 1.0f);
  float fd = 3e38f;
  bool alive = false;
  {
    /* faces (0,1,2), (0,3,1), (0,2,3), (1,3,2) and the face across each of their edges; all four are turned when face 0 looks at vertex 3 */
    const V3 q0 = GQ_CVX_PW(P, 0), q1 = GQ_CVX_PW(P, 1), q2 = GQ_CVX_PW(P, 2), q3 = GQ_CVX_PW(P, 3);
    V3 n0; float dd0;
    cvx_face_plane(q0, q1, q2, n0, dd0);
    const bool turn = dot(n0, q3) - dd0 > 0.0f;
    const int l4 = lane & 3;
    const int t0 = l4 == 3 ? 1 : 0, t1 = l4 == 0 ? 1 : (l4 == 2 ? 2 : 3), t2 = l4 == 0 ? 2 : (l4 == 1 ? 1 : (l4 == 2 ? 3 : 2));
    const int a0 = l4 == 0 ? 1 : (l4 == 1 ? 2 : (l4 == 2 ? 0 : 1)), a1 = l4 == 3 ? 2 : 3, a2 = l4 == 0 ? 2 : (l4 == 1 ? 0 : (l4 == 2 ? 1 : 0));
    const int u1 = turn ? t2 : t1, u2 = turn ? t1 : t2;
    fv = t0 | (u1 << 8) | (u2 << 16);
    alive = lane < 4;
    cvx_face_plane(GQ_CVX_PW(P, t0), GQ_CVX_PW(P, u1), GQ_CVX_PW(P, u2), fn, fd);
    wave_barrier();
    ADJ[lane] = lane < 4 ? ((turn ? a2 : a0) | (a1 << 8) | ((turn ? a0 : a2) << 16)) : 0;
    wave_barrier();
  }
  int nv = 4, best = 0;
#pragma unroll 1
  for (int eit = 0;; eit++) {
    const float dmin = wave_min(alive ? fd : 3e38f);
    const uint64_t bm = ballot(alive && fd == dmin);
    if (bm == 0) return false;
    best = ffs64(bm);
    if (eit >= GQ_CVX_EPA_MAXIT || nv >= GQ_CVX_MAXV) break;
    const V3 nb = v3(bcast(fn.x, best), bcast(fn.y, best), bcast(fn.z, best));
    GQ_CVX_T(3);
    const CvxMink sw = cvx_minkowski(G, vx, vy, vz, nb, caps);
    GQ_CVX_T(1);
    const V3 w = sw.w;
    const int wid = sw.id;
    const bool dup = ballot(lane < nv && GQ_CVX_PID(P, lane < nv ? lane : 0) == wid) != 0;
#ifdef GQ_EMU_TRACE
    if (lane == 0 && getenv("GQ_CVX_TRACE")) printf("epa it %d best %d dmin %.9g sup %.9g dup %d nv %d wid %x\n", eit, best, (double)dmin, (double)dot(w, nb), (int)dup, nv, wid);
#endif
    if (dup || dot(w, nb) - dmin <= GQ_CVX_TOL_EPA) break; /* the face lies on the boundary of A - B */
    /* the connected patch of faces that see w, grown from the closest one through shared edges */
    const bool cand = alive && fd < 1e29f && dot(fn, w) - fd > 0.5f * GQ_CVX_TOL_EPA;
    const int adj = ADJ[lane];
    const int j0 = adj & 0xff, j1 = (adj >> 8) & 0xff, j2 = (adj >> 16) & 0xff;
    uint64_t vis = 1ull << best;
    for (;;) {
      const bool in = cand && !((vis >> lane) & 1ull) && (((vis >> j0) | (vis >> j1) | (vis >> j2)) & 1ull);
      const uint64_t add = ballot(in);
      if (!add) break;
      vis |= add;
    }
    const bool mine = (vis >> lane) & 1ull;
    const bool r0 = mine && !((vis >> j0) & 1ull), r1 = mine && !((vis >> j1) & 1ull), r2 = mine && !((vis >> j2) & 1ull);
    const int cnt = (r0 ? 1 : 0) + (r1 ? 1 : 0) + (r2 ? 1 : 0);
    const int incl = wave_incl_scan(cnt);
    const int nh = bcast(incl, 63);
    if (nh > GQ_CVX_MAXRIM) break; /* (a rim of more than 24 edges: the polytope has outgrown the scratch; the closest face so far is the answer) */
    {
      int at = incl - cnt;
      const int f0 = fv & 0xff, f1 = (fv >> 8) & 0xff, f2 = (fv >> 16) & 0xff;
      if (r0) RIM[at++] = f0 | (f1 << 8) | (j0 << 16);
      if (r1) RIM[at++] = f1 | (f2 << 8) | (j1 << 16);
      if (r2) RIM[at++] = f2 | (f0 << 8) | (j2 << 16);
    }
    alive = alive && !mine;
    st3l(P + 4 * nv, w); PI[4 * nv + 3] = wid; /* (every lane: the same words) */
    wave_barrier();
    GQ_CVX_T(4);
    /* the fan: new face k = rim edge k + the new vertex, on the k-th free lane */
    const uint64_t freem = ballot(!alive);
    const int rank = popc64(freem & lt);
    const bool take = !alive && rank < nh;
    int ea = 0, eb = 0, eg = 0;
    if (take) {
      const int e = RIM[rank];
      ea = e & 0xff; eb = (e >> 8) & 0xff; eg = (e >> 16) & 0xff;
    }
    const int gfv = shfl_idx(fv, take ? eg : lane); /* the face behind the rim edge: its vertices (it is not in the patch: its lane keeps them) */
    /* the fan's own neighbours: the new face whose rim edge starts where this one's ends, and the one whose edge ends where this one's
     * starts - first match in rim order = lane order of the taking lanes (v_readlane per rim edge: the loop over the LDS list it replaces
     * was a dependent LDS read per edge) */
    int n1 = lane, n2 = lane;
    {
      bool s1 = false, s2 = false;
      for (uint64_t tm = ballot(take); tm;) { /* wave-uniform */
        const int src = ffs64(tm); tm &= tm - 1;
        const int ka = bcast(ea, src), kb = bcast(eb, src);
        if (!s1 && ka == eb) { n1 = src; s1 = true; }
        if (!s2 && kb == ea) { n2 = src; s2 = true; }
      }
    }
    wave_barrier();
    if (take) {
      fv = ea | (eb << 8) | (nv << 16);
      cvx_face_plane(GQ_CVX_PW(P, ea), GQ_CVX_PW(P, eb), w, fn, fd);
      alive = true;
      ADJ[lane] = eg | (n1 << 8) | (n2 << 16);
      /* the face behind the rim edge now borders this one: its edge b -> a */
      const int g0 = gfv & 0xff, g1 = (gfv >> 8) & 0xff, g2 = (gfv >> 16) & 0xff;
      const int q = (g0 == eb && g1 == ea) ? 0 : ((g1 == eb && g2 == ea) ? 1 : 2);
      ((GQ_LDS uint8_t*)ADJ)[4 * eg + q] = (uint8_t)lane;
    }
    nv++;
    wave_barrier();
    GQ_CVX_T(5);
  }
  GQ_CVX_T(3);
#ifdef GQ_CVX_STATS
  ((LdsI)out)[7] = nv - 4;
#endif
  /* the closest face: the foot point of its plane, split over the face's vertices.  A facet of A - B with more than three vertices (edge
   * against edge: a parallelogram) is several coplanar triangles of equal offset - the one that CONTAINS the foot point carries the
   * witness points: among the faces within the tolerance of the smallest offset (lane = face), the one whose nearest point is nearest */
  {
    const float dsel = wave_min(alive ? fd : 3e38f);
    float q2 = 3e38f;
    if (alive && fd <= dsel + 10.0f * GQ_CVX_TOL_EPA) {
      const D3 a = d3(GQ_CVX_PW(P, fv & 0xff)), b = d3(GQ_CVX_PW(P, (fv >> 8) & 0xff)), c = d3(GQ_CVX_PW(P, (fv >> 16) & 0xff));
      double l3[3];
      cvx_tri(a, b, c, l3);
      const V3 q = cvx_comb(l3, a, b, c);
      q2 = dot(q, q);
    }
    const float q2min = wave_min(q2);
    const uint64_t sel = ballot(q2 == q2min && q2 < 3e38f);
    if (sel) best = ffs64(sel);
  }
  {
    const int bf = bcast(fv, best);
    const V3 n = v3(bcast(fn.x, best), bcast(fn.y, best), bcast(fn.z, best));
    const float dbest = bcast(fd, best);
    const float dd = dbest < 1e29f ? dbest : 0.0f;
    const int i0 = bf & 0xff, i1 = (bf >> 8) & 0xff, i2 = (bf >> 16) & 0xff;
    double l3[3];
    const D3 foot = d3(dd * n);
    cvx_tri(d3(GQ_CVX_PW(P, i0)) - foot, d3(GQ_CVX_PW(P, i1)) - foot, d3(GQ_CVX_PW(P, i2)) - foot, l3);
    const V3 pa = (float)l3[0] * cvx_point(SA, vx, vy, vz, GQ_CVX_PID(P, i0) & 0xffff) + (float)l3[1] * cvx_point(SA, vx, vy, vz, GQ_CVX_PID(P, i1) & 0xffff) +
                  (float)l3[2] * cvx_point(SA, vx, vy, vz, GQ_CVX_PID(P, i2) & 0xffff);
    const V3 pb = pa - dd * n;
    const float dist = -dd - rA - rB;
    if (!(dist < margin)) return false;
    wave_barrier();
    out[0] = dist; st3l(out + 1, n); st3l(out + 4, 0.5f * ((pa + rA * n) + (pb - rB * n)));
    wave_barrier();
    return true;
  }
}

/* An UPPER bound of a cloud's support function h(d) = max over its vertices v of v . d, d in the geom frame (any length), ONE lane: the
 * bilinear blend of the four nodes of d's cell in the cloud's cube-map table (GqModelDesc.support_grid, 6 x 17 x 17 values at vx[adr..]) -
 * h is convex and positively homogeneous, the blend of a cell's corners bounds it from above, second order in the cell size.  With it a lane
 * proves a pair of hulls apart along a direction - no contact within the margin - without the routine's wave-wide support query. */
__device__ inline float cvx_hgrid(const GQ_MODEL float* vx, int adr, V3 d) {
  const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
  int face; float mj, a, b;
  if (ax >= ay && ax >= az) { face = d.x >= 0.0f ? 0 : 1; mj = ax; a = d.y; b = d.z; }
  else if (ay >= az) { face = d.y >= 0.0f ? 2 : 3; mj = ay; a = d.x; b = d.z; }
  else { face = d.z >= 0.0f ? 4 : 5; mj = az; a = d.x; b = d.y; }
  if (!(mj > 1e-30f)) return 3e38f;
  const float inv = fast_rcp(mj);
  constexpr int NG = GQ_SUPPORT_GRID, NN = NG + 1;
  const float fa = fminf(fmaxf((a * inv + 1.0f) * (0.5f * NG), 0.0f), (float)NG), fb = fminf(fmaxf((b * inv + 1.0f) * (0.5f * NG), 0.0f), (float)NG);
  const int ia = imin((int)fa, NG - 1), ib = imin((int)fb, NG - 1);
  const float ta = fa - (float)ia, tb = fb - (float)ib;
  const GQ_MODEL float* T = vx + adr + face * (NN * NN) + ia * NN + ib;
  const float h0 = T[0] + tb * (T[1] - T[0]), h1 = T[NN] + tb * (T[NN + 1] - T[NN]);
  return mj * (h0 + ta * (h1 - h0)) + 2e-6f * mj; /* (fp32 blend: a hair on top) */
}

/* mid phase of a convex pair, ONE lane: two oriented boxes (centre, axes = columns of R, half extents) are held apart by more than
 * `reach` along one of the 15 separating-axis candidates (oracle obb_apart) - then so are the hulls inside them */
__device__ inline bool obb_apart(V3 ca, const float* Ra, V3 ha, V3 cb, const float* Rb, V3 hb, float reach) {
  float C[3][3], AC[3][3], t[3];
  const V3 d = cb - ca;
  const float hA[3] = {ha.x, ha.y, ha.z}, hB[3] = {hb.x, hb.y, hb.z};
#pragma unroll
  for (int i = 0; i < 3; i++) {
    t[i] = Ra[i] * d.x + Ra[3 + i] * d.y + Ra[6 + i] * d.z;
#pragma unroll
    for (int j = 0; j < 3; j++) { C[i][j] = Ra[i] * Rb[j] + Ra[3 + i] * Rb[3 + j] + Ra[6 + i] * Rb[6 + j]; AC[i][j] = fabsf(C[i][j]) + 1e-6f; }
  }
  bool apart = false;
#pragma unroll
  for (int i = 0; i < 3; i++) apart = apart || fabsf(t[i]) - (hA[i] + hB[0] * AC[i][0] + hB[1] * AC[i][1] + hB[2] * AC[i][2]) > reach;
#pragma unroll
  for (int j = 0; j < 3; j++) apart = apart || fabsf(t[0] * C[0][j] + t[1] * C[1][j] + t[2] * C[2][j]) - (hB[j] + hA[0] * AC[0][j] + hA[1] * AC[1][j] + hA[2] * AC[2][j]) > reach;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      const float len2 = 1.0f - C[i][j] * C[i][j];
      const float sep = fabsf(t[i2] * C[i1][j] - t[i1] * C[i2][j]) - (hA[i1] * AC[i2][j] + hA[i2] * AC[i1][j] + hB[j1] * AC[i][j2] + hB[j2] * AC[i][j1]);
      apart = apart || (len2 >= 1e-4f && sep > reach * fast_sqrt(len2));
    }
  return apart;
}

}  // namespace gq
