/*
 * gq_step_body.h - body of the fused step kernel (see gq_step_kernel.h for the stage map and the reference
 * lines each stage replaces).  Everything runs in one wavefront; `wave_barrier()` separates producer and
 * consumer lanes of an LDS hand-off.
 */
#pragma once
#include "gq_step_kernel.h"
#include "gq_newton.h"
#include "gq_boxes.h"
#include "gq_heightmap.h"

namespace gq {

/* entry (i,j) of the joint-space inertia from its tree-sparse storage */
__device__ __forceinline__ float m_entry(const WaveMem& W, int i, int j) {
  if (i < j) { int t = i; i = j; j = t; }
  if (i < 6) return W.Mb[i][j];
  if (j < 6) return W.Mc[i - 6][j];
  if ((i - 6) / 3 != (j - 6) / 3) return 0.0f;
  return W.Mc[i - 6][6 + (j - 6) % 3];
}

/* S1 (mj_kinematics): lanes 0-3 walk the leg chains; returns the normalised base quaternion (all lanes) */
__device__ inline Q4 stage_kinematics(WaveMem& W, const GqDevLinkRec L) { /* (by value: the callee is out of line, a reference would put the record in scratch memory) */
  const int lane = lane_id();
  Q4 qbase = {W.qb[0], W.qb[1], W.qb[2], W.qb[3]};
  qbase = qnormalize(qbase);
  { /* (every lane: same words, same values) */
    W.xpos[0][0] = 0.0f; W.xpos[0][1] = 0.0f; W.xpos[0][2] = W.basez;
    q2mat(W.xmat[0], qbase);
  }
  /* phase 1, lane = link (12 lanes): local transform of the link in its parent's frame - all model reads and the
   * sin/cos of the joint angle happen here, in parallel */
  { /* lane = link, mirror lanes (the link's record - body_quat, joint axis / anchor, and the state-independent anchor and axis in the parent
     * frame, folded on the host - came in one batch: no model read in front of the sin/cos) */
    const int j = lane < GQ_NJ ? lane : GQ_NJ - 1;
    const Q4 bq = {L.bq[0], L.bq[1], L.bq[2], L.bq[3]};
    const V3 jp = ld3(L.jp), ax = ld3(L.ax);
    float R1[9], sn, cs;
    sincos_small(0.5f * (W.qj[j] - L.qpos0), sn, cs);
    const Q4 qr = {cs, ax.x * sn, ax.y * sn, ax.z * sn};
    const Q4 ql = qmul(bq, qr);
    q2mat(R1, ql);
    const V3 aloc = ld3(L.aloc);                    /* joint anchor in the parent frame */
    const V3 ploc = aloc - matvec(R1, jp);          /* child origin: rotation about the anchor keeps it fixed */
    float* o = W.u.dyn.fkloc[j];
    o[0] = ql.w; o[1] = ql.x; o[2] = ql.y; o[3] = ql.z;
    st3(o + 4, ploc); st3(o + 7, aloc); st3(o + 10, ld3(L.r0ax));
  }
  wave_barrier();
  GQ_SUB(W, 1, 2); /* kinematics phase 1 */
  /* phase 2, lane = leg: compose the three local transforms down the chain */
  { /* lane = leg, mirror lanes */
    const int leg = lane < 4 ? lane : 3;
    float Rp[9];
    q2mat(Rp, qbase);
    V3 pp = v3(0.0f, 0.0f, W.basez);
    Q4 pq = qbase;
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const int b = 1 + 3 * leg + i, j = 3 * leg + i;
      const float* o = W.u.dyn.fkloc[j];
      const Q4 ql = {o[0], o[1], o[2], o[3]};
      st3(W.u.dyn.anchor[j], pp + matvec(Rp, ld3(o + 7)));   /* anchor/axis do not overlay fkloc */
      st3(W.u.dyn.axis[j], matvec(Rp, ld3(o + 10)));
      const V3 pos = pp + matvec(Rp, ld3(o + 4));
      const Q4 q = qnormalize(qmul(pq, ql));
      q2mat(Rp, q);
      st3(W.xpos[b], pos);
#pragma unroll
      for (int k = 0; k < 9; k++) W.xmat[b][k] = Rp[k];
      pp = pos; pq = q;
    }
  }
  wave_barrier();

  return qbase;
}

/* S6 (mj_collision, floor plane z = 0): foot sphere centres and, per link geom, the deepest cloud vertex.
 * calf_only restricts the scan to geoms of the calf bodies (reset lift loop, quadruped_env.py:376-388). */
/* floor pass scratch in the J block (idle between S5 and the world-box / self-collision passes of S6): per link geom up to two further
 * contact points of a mesh with the floor (geom-frame point, distance; 1e30: none) and the support vertex's index in its cloud */
#define GQ_FLR_XTRA(W) (&(W).u.B[0][0])                                   /* [GQ_MAXLG][2][4] */
#define GQ_FLR_WIDX(W) (reinterpret_cast<int32_t*>(&(W).u.B[17][0]))      /* [GQ_MAXLG] */
static_assert(8 * GQ_MAXLG <= 17 * GQ_NVD && GQ_MAXLG <= 3 * GQ_NVD, "floor pass scratch");
struct FootRec { int leg; float pos[3]; }; /* lane = foot: its leg and the sphere centre in the calf frame (fetched a stage early by the caller) */
template <class M> __device__ __forceinline__ FootRec foot_fetch(const M& m, const int lane) {
  const int k = lane < 4 ? lane : 3;
  FootRec r;
  r.leg = m.foot_leg[k]; r.pos[0] = m.foot_pos[k][0]; r.pos[1] = m.foot_pos[k][1]; r.pos[2] = m.foot_pos[k][2];
  return r;
}
__device__ inline void stage_collision_scan(WaveMem& W, const GQ_MODEL GqDevModel& m, const GQ_MODEL float* vx, const GQ_MODEL float* vy, const GQ_MODEL float* vz,
                                            bool calf_only, const FootRec FR) {
  const int lane = lane_id();
  { /* feet: exact plane-sphere; lane = foot, mirror lanes (the record is fetched with the clamped lane) */
    const int b = 3 + 3 * FR.leg;
    V3 c = ld3(W.xpos[b]) + matvec(W.xmat[b], ld3(FR.pos));
    st3(W.foot_world[lane < 4 ? lane : 3], c);
  }
  /* link geoms.  Phase 1, lane = geom: plane normal in the geom frame, the OBB lower bound of the cloud, and - for a geom that may
   * touch - the mask of the 64-vertex chunks of its DIRECTION-ordered cloud that can hold the support vertex of that normal (one word
   * of the geom's cube-map table).
   * Phase 3, wave-uniform loop over the surviving geoms: wave-wide scan of the masked chunks (two or three of up to eleven) for the
   * deepest vertex.
   * Phase 4, lane = geom again: the winner's world position.
   * The scan is a chain of memory round trips - a robot lying on the floor has ten geoms to scan and is also the env whose
   * Newton solve ends the launch - so a geom's table addresses come from its phase-1 lane (ds_bpermute / v_readlane), not
   * from scalar loads in front of the vertex loads; every lane keeps the COORDINATES of its best vertex, so the winner needs no
   * second, dependent fetch; the model reads of the final transform happen once, for all geoms together. */
  const int nlg = m.nlg;
  V3 ng = v3(0.0f, 0.0f, 0.0f);
  float d0 = 0.0f, gmargin = 0.0f, gradius = 0.0f;
  bool needs = false;
  int cadr = 0, cnum = 0, cmask0 = 0;
  if (nlg > 0) { /* wave-uniform; lane = geom, mirror lanes (they never join the scan: `needs` is masked) */
    const int lg_ = lane < nlg ? lane : nlg - 1;
    const GQ_MODEL GqDevGeom& G = m.lg[lg_];
    const float* Rb = W.xmat[G.body];
    /* plane normal in the geom frame: n_g = Rg' Rb' n, n = (0,0,1) */
    const V3 nb = v3(Rb[6], Rb[7], Rb[8]);
    ng = matTvec(G.mat, nb);
    d0 = W.xpos[G.body][2] + dot(nb, ld3(G.pos));
    gmargin = G.margin; gradius = G.radius;
    const float lower = d0 + dot(ng, ld3(G.aabb_c)) - (fabsf(ng.x) * G.aabb_h[0] + fabsf(ng.y) * G.aabb_h[1] + fabsf(ng.z) * G.aabb_h[2]) - gradius;
    const bool calf = G.body > 0 && (G.body - 1) % 3 == 2;
    needs = lane < nlg && G.ptype == 0 && lower < gmargin && (!calf_only || calf); /* primitive geoms are evaluated lane-locally (floor_candidates) */
    cadr = G.plane_adr; cnum = G.cloud_num; /* the DIRECTION-ordered copy of the cloud */
    { /* which chunks can hold the support vertex of direction -n_g: the geom's mask table, one word per cube-map cell of the direction
       * (host: cabi.plane_support_tables; no table: every chunk) */
      const int pm = G.pmask_adr, gridn = m.plane_grid;
      const float ax_ = fabsf(ng.x), ay_ = fabsf(ng.y), az_ = fabsf(ng.z);
      const int mx_ = (ax_ >= ay_ && ax_ >= az_) ? 0 : (ay_ >= az_ ? 1 : 2);
      const float dm = -(mx_ == 0 ? ng.x : (mx_ == 1 ? ng.y : ng.z));
      const float o0 = -(mx_ == 0 ? ng.y : ng.x), o1 = -(mx_ == 2 ? ng.y : ng.z);
      const float inv = fast_rcp(fmaxf(fabsf(dm), 0.5f)); /* (the dominant component of a unit vector is >= 0.577) */
      const float hg = 0.5f * (float)gridn;
      const int iu = imin(imax((int)((o0 * inv + 1.0f) * hg), 0), gridn - 1), iv = imin(imax((int)((o1 * inv + 1.0f) * hg), 0), gridn - 1);
      const int cell = ((mx_ * 2 + (dm > 0.0f ? 0 : 1)) * gridn + iu) * gridn + iv;
      const float mw = vx[pm >= 0 ? pm + cell : 0]; /* (unconditional load, clamped address) */
      const int nchunk = (cnum + GQ_WAVE - 1) / GQ_WAVE;
      cmask0 = pm >= 0 ? (int)mw : ((1 << nchunk) - 1);
    }
    W.u2.c.lg_dist[lg_] = 1e30f; /* (a geom that is scanned gets its distance from phase 3) */
    GQ_FLR_XTRA(W)[8 * lg_ + 3] = 1e30f; GQ_FLR_XTRA(W)[8 * lg_ + 7] = 1e30f; /* (... and its further contact points from phase 5) */
  }
  uint64_t todo = ballot(needs);
  if (todo) { /* wave-uniform */
    /* (phase 2 - chunk masks from the boxes of position-sorted chunks, one memory round trip per four geoms - is gone: the masks come
     * from the direction table in phase 1.  A link lying on the floor has every slab along its axis equally deep: all 8 - 11 chunks of
     * each of its ten geoms were scanned, 18 k cycles for exactly the waves that end a launch; by direction it is two or three) */
    const int cmask = needs ? cmask0 : 0;
    /* phase 3 */
    float px[4], py[4], pz[4];
    for (;;) { /* one trip per geom */
      const int g = ffs64(todo);
      todo &= todo - 1;
      const float gx = bcast(ng.x, g), gy = bcast(ng.y, g), gz = bcast(ng.z, g);
      const int adr = bcast(cadr, g), last = adr + bcast(cnum, g) - 1;
      int cm = bcast(cmask, g);
      float best = 1e30f;
      V3 bp = v3(0.0f, 0.0f, 0.0f);
      int bi = 0;
      while (cm) { /* wave-uniform: up to 4 chunks with all 12 loads in flight together; lanes past the end re-read the last vertex */
        int cu[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { cu[u] = cm ? __builtin_ctz(cm) : -1; cm &= cm - 1; }
#pragma unroll
        for (int u = 0; u < 4; u++)
          if (cu[u] >= 0) { /* wave-uniform */
            const int i = adr + cu[u] * GQ_WAVE + lane;
            const int ii = i < last ? i : last;
            px[u] = vx[ii]; py[u] = vy[ii]; pz[u] = vz[ii];
          }
#pragma unroll
        for (int u = 0; u < 4; u++)
          if (cu[u] >= 0) {
            const float dv = gx * px[u] + gy * py[u] + gz * pz[u];
            const int i = adr + cu[u] * GQ_WAVE + lane;
            if (dv < best) { best = dv; bp = v3(px[u], py[u], pz[u]); bi = (i < last ? i : last) - adr; } /* (a re-read last vertex never wins a tie: strict <) */
          }
      }
      const float wmin = wave_min(best);
      const int who = ffs64(ballot(best == wmin));
      if (lane == who) { W.u2.c.lg_dist[g] = wmin; st3(W.u2.c.lg_pt[g], bp); GQ_FLR_WIDX(W)[g] = bi; } /* geom-frame vertex for now; 1e30: no chunk in reach */
      if (!todo) break;
    }
    wave_barrier();
    int nb_adr = 0, nb_cnt = 0;
    if (needs) { /* phase 4, lane = geom */
      const GQ_MODEL GqDevGeom& G = m.lg[lane];
      const float* Rb = W.xmat[G.body];
      const float raw = W.u2.c.lg_dist[lane];
      const V3 vb = ld3(G.pos) + matvec(G.mat, ld3(W.u2.c.lg_pt[lane]));
      const float dist = raw < 1e29f ? raw + d0 - gradius : 1e30f;
      W.u2.c.lg_dist[lane] = dist;
      st3(W.u2.c.lg_pt[lane], ld3(W.xpos[G.body]) + matvec(Rb, vb));
      /* mjc_PlaneConvex's neighbour walk: a mesh that touches brings the hull-graph neighbours of its support vertex (the record of the
       * winner: first entry and length of its list in the vertex arrays) */
      if (dist < gmargin && G.nbr_adr >= 0) {
        const int r = G.nbr_adr + GQ_FLR_WIDX(W)[lane];
        nb_adr = (int)vx[r]; nb_cnt = (int)vy[r];
      }
    }
    /* phase 5, one trip per touching mesh, lane = neighbour (list order = ascending vertex index of the hull table): the first two inside
     * the margin join the support vertex - geom-frame coordinates and distance go to GQ_FLR_XTRA, floor_candidates takes them to the world */
    uint64_t walk = ballot(nb_cnt > 0);
    while (walk) { /* wave-uniform */
      const int g = ffs64(walk);
      walk &= walk - 1;
      const int a0 = bcast(nb_adr, g), cnt = imin(bcast(nb_cnt, g), GQ_WAVE);
      const float gx = bcast(ng.x, g), gy = bcast(ng.y, g), gz = bcast(ng.z, g), off = bcast(d0 - gradius, g), mg = bcast(gmargin, g);
      const int i = a0 + (lane < cnt ? lane : 0);
      const V3 p = v3(vx[i], vy[i], vz[i]);
      const float dk = gx * p.x + gy * p.y + gz * p.z + off;
      uint64_t in = ballot(lane < cnt && dk < mg);
      float* X = GQ_FLR_XTRA(W) + 8 * g;
#pragma unroll
      for (int q = 0; q < 2; q++) {
        const int who = in ? ffs64(in) : -1;
        in &= in - 1;
        if (lane == who) { st3(X + 4 * q, p); X[4 * q + 3] = dk; }
        if (who < 0 && lane == 0) X[4 * q + 3] = 1e30f;
      }
    }
  }
  wave_barrier();
}

/* Candidate contact points of collision item `code` (k < 4: foot k, else 4 + link geom) with the floor plane z = 0, evaluated
 * by ONE lane - MuJoCo's plane routines per geom type (restated in oracle/gq_oracle.c::gqo_collision):
 *   sphere    mjraw_PlaneSphere   one point
 *   capsule   mjraw_PlaneCapsule  both end spheres, the +axis end first; first tangent of the frame = the capsule axis made
 *                                 orthogonal to the normal
 *   box       mjraw_PlaneBox      the corners at or below the box centre, in corner order, at most 4
 *   cylinder  mjc_PlaneCylinder   lowest rim point of the near cap, the rim point under it on the far cap, two more points of
 *                                 the near cap at +-120 degrees
 *   hull      mjc_PlaneConvex     the support vertex (found by the 64-lane scan of stage_collision_scan)
 * n candidates in MuJoCo's order: reference point pt[k] (sphere centre / corner / rim point) and distance dist[k] = pt.z - r.
 * Whether a candidate IS a contact is its own `dist < margin` test - equivalent to the routines' early exits, because the first
 * candidate of a cylinder is never farther than the others.  A lift by dz moves every pt.z and dist by dz. */
struct FloorCand { int n; float r, t1c, t1s; float dist[4]; V3 pt[4]; int key; }; /* key: 2 bits per candidate = its place in MuJoCo's order */
__device__ inline void floor_candidates(const WaveMem& W, const ItemRegs& G, FloorCand& C) {
  C.n = 0; C.r = 0.0f; C.t1c = 0.0f; C.t1s = 1.0f; C.key = 0xE4; /* places 0, 1, 2, 3 */
#pragma unroll
  for (int k = 0; k < 4; k++) { C.dist[k] = 1e30f; C.pt[k] = v3(0.0f, 0.0f, 0.0f); }
  const int ptype = G.ptype;
  if (ptype < 0) { /* foot sphere */
    C.n = 1; C.r = G.radius; C.pt[0] = ld3(W.foot_world[G.code]); C.dist[0] = C.pt[0].z - C.r;
    return;
  }
  if (ptype == 0) { /* hull: the support vertex, then up to two of its hull-graph neighbours (stage_collision_scan phase 5) */
    const int g = G.code - 4;
    C.n = 1; C.r = G.radius; C.pt[0] = ld3(W.u2.c.lg_pt[g]); C.dist[0] = W.u2.c.lg_dist[g];
    const float* X = GQ_FLR_XTRA(W) + 8 * g;
    const float d1 = X[3], d2 = X[7];
    if (d1 < 1e29f) { /* (rare: only a mesh that touches has them) */
      const float* Rb = W.xmat[G.body];
      const V3 o = ld3(W.xpos[G.body]) + matvec(Rb, G.pos);
      C.pt[1] = o + matvec(Rb, matvec(G.mat, ld3(X))); C.dist[1] = d1; C.n = 2;
      if (d2 < 1e29f) { C.pt[2] = o + matvec(Rb, matvec(G.mat, ld3(X + 4))); C.dist[2] = d2; C.n = 3; }
    }
    return;
  }
  const float* Rb = W.xmat[G.body];
  const V3 c = ld3(W.xpos[G.body]) + matvec(Rb, G.pos);
  float A[9]; /* geom frame in the world: Rb Rg */
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) A[3 * i + j] = Rb[3 * i] * G.mat[j] + Rb[3 * i + 1] * G.mat[3 + j] + Rb[3 * i + 2] * G.mat[6 + j];
  if (ptype == 2) { C.n = 1; C.r = G.psize[0]; C.pt[0] = c; C.dist[0] = c.z - C.r; }
  else if (ptype == 3) {
    const V3 ax = v3(A[2], A[5], A[8]);
    const float hl = G.psize[1];
    C.n = 2; C.r = G.psize[0];
    C.pt[0] = c + hl * ax; C.pt[1] = c - hl * ax;
    C.dist[0] = C.pt[0].z - C.r; C.dist[1] = C.pt[1].z - C.r;
    const float l2 = ax.x * ax.x + ax.y * ax.y;
    if (l2 > 1e-30f) { const float inv = fast_rsqrt(l2); C.t1c = ax.x * inv; C.t1s = ax.y * inv; }
    else { C.t1c = 1.0f; C.t1s = 0.0f; } /* mju_normalize3 of a null vector */
  } else if (ptype == 6) {
    /* corner i = centre + s0 ux + s1 uy + s2 uz, sign s_b = bit b of i.  Corners i and 7 - i are opposite (ldist_i = -ldist_(7-i)):
     * of each of the four pairs (i, 7 - i), i < 4, exactly one lies at or below the centre - so the candidates are one corner
     * per pair, kept in PAIR order in the registers; MuJoCo walks the corners by index and keeps the first four that qualify,
     * i.e. the chosen i < 4 ascending, then the chosen 7 - i ascending: that place goes into `key`, and the contact list ranks
     * the touching candidates by it (no data movement).  (An exact tie ldist_i = 0 makes both corners of a pair qualify in
     * MuJoCo; here the lower index is taken - measure zero.) */
    const V3 ux = G.psize[0] * v3(A[0], A[3], A[6]), uy = G.psize[1] * v3(A[1], A[4], A[7]), uz = G.psize[2] * v3(A[2], A[5], A[8]);
    bool low[4];
    int nlow = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const V3 off = ((i & 1) ? ux : -1.0f * ux) + ((i & 2) ? uy : -1.0f * uy) - uz;
      low[i] = off.z <= 0.0f;
      const V3 o = low[i] ? off : -1.0f * off;
      C.pt[i] = c + o; C.dist[i] = c.z + o.z;
      nlow += low[i] ? 1 : 0;
    }
    int key = 0, seen = 0, after = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) { if (low[i]) { key |= seen << (2 * i); seen++; } }
#pragma unroll
    for (int i = 3; i >= 0; i--) { if (!low[i]) { key |= (nlow + after) << (2 * i); after++; } } /* 7 - i ascending = i descending */
    C.key = key;
    C.n = 4;
  } else if (ptype == 5) {
    V3 ax = v3(A[2], A[5], A[8]);
    float prjaxis = ax.z;
    if (prjaxis > 0.0f) { ax = -1.0f * ax; prjaxis = -prjaxis; } /* the axis points towards the plane */
    const float rad = G.psize[0], hl = G.psize[1];
    V3 vec = prjaxis * ax - v3(0.0f, 0.0f, 1.0f); /* -normal without its component along the axis */
    const float len2 = dot(vec, vec);
    if (len2 >= 1e-30f) vec = (rad * fast_rsqrt(len2)) * vec;
    else vec = rad * v3(A[0], A[3], A[6]); /* disk parallel to the plane: the cylinder's x axis */
    const float prjvec = vec.z;
    const V3 axs = hl * ax;
    prjaxis *= hl;
    C.n = 4;
    C.pt[0] = c + vec + axs; C.dist[0] = c.z + prjaxis + prjvec;
    C.pt[1] = c + vec - axs; C.dist[1] = c.z - prjaxis + prjvec;
    V3 v1 = cross(vec, axs);
    const float n1 = dot(v1, v1);
    v1 = n1 > 1e-30f ? (rad * 0.8660254037844386f * fast_rsqrt(n1)) * v1 : v3(rad * 0.8660254037844386f, 0.0f, 0.0f);
    C.pt[2] = c + v1 + axs - 0.5f * vec; C.pt[3] = c - v1 + axs - 0.5f * vec;
    C.dist[2] = C.dist[3] = c.z + prjaxis - 0.5f * prjvec;
  }
}

/* _sample_ref_vel (quadruped_env.py:1046-1072) and _sample_external_disturbances (:1074-1139) for the wave's env when their
 * step countdowns run out (:292-305).  Counter-based draws: Philox block (b, n, global env id, 0xc0de | 0xd157), n = number
 * of redraws so far; uniform = (word >> 8) 2^-24; np.random.randint(1000, 3000) = 1000 + floor(2000 u). */
template <bool PUB = false>
__device__ inline void resample_wave(const StepArgs& a, const int env) {
  const int lane = lane_id();
  const GQ_MODEL GqDevBatch& B = *mptr(a.batch);
  GQ_GLOBAL int32_t* hc = gptr(a.h9) + (size_t)env * 6;
  const uint32_t gid = (uint32_t)(env + B.rs_env_id_offset);
  if (B.rs_cmd_reset) {
    const int after = ldv<PUB>(a.h9 + (size_t)env * 6 + 0) + 1, before = ldv<PUB>(a.h9 + (size_t)env * 6 + 1); /* wave-uniform */
    if (after >= before) {
      const int n = ldv<PUB>(a.h9 + (size_t)env * 6 + 2);
      float u = 0.0f;
      if (lane < 4) u = (float)(philox4x32(0u, (uint32_t)n, gid, 0xc0deu, B.rs_seed_lo, B.rs_seed_hi, lane) >> 8) * (1.0f / 16777216.0f);
      const float u_norm = bcast(u, 0), u_head = bcast(u, 1), u_yaw = bcast(u, 2), u_int = bcast(u, 3);
      if (lane == 0) {
        float norm = 0.0f, heading = 0.0f, yaw_dot = 0.0f;
        if (B.rs_cmd_forward) norm = B.rs_lin_vel_range[0] + (B.rs_lin_vel_range[1] - B.rs_lin_vel_range[0]) * u_norm;
        else if (B.rs_cmd_random) {
          norm = B.rs_lin_vel_range[0] + (B.rs_lin_vel_range[1] - B.rs_lin_vel_range[0]) * u_norm;
          heading = (2.0f * u_head - 1.0f) * 3.14159265358979f;
        }
        if (B.rs_cmd_rotate) yaw_dot = B.rs_ang_vel_range[0] + (B.rs_ang_vel_range[1] - B.rs_ang_vel_range[0]) * u_yaw;
        GQ_GLOBAL float* cmd = gptr(const_cast<float*>(a.cmd)) + (size_t)env * 4;
        float sh, ch;
        sincos_small(heading, sh, ch);
        cmd[0] = norm * ch; cmd[1] = norm * sh; cmd[2] = 0.0f; cmd[3] = yaw_dot;
        hc[0] = 0; hc[1] = 1000 + (int)(2000.0f * u_int); hc[2] = n + 1;
      }
    } else if (lane == 0) hc[0] = after;
  }
  if (B.rs_dist_reset && a.ext_dist) {
    const int after = ldv<PUB>(a.h9 + (size_t)env * 6 + 3) + 1, before = ldv<PUB>(a.h9 + (size_t)env * 6 + 4);
    GQ_GLOBAL float* ed = gptr(a.ext_dist) + (size_t)env * 6;
    float val = lane < 6 ? ldv<PUB>(a.ext_dist + (size_t)env * 6 + lane) : 0.0f;
    if (after >= before) {
      const int n = ldv<PUB>(a.h9 + (size_t)env * 6 + 5);
      float u = 0.0f;
      if (lane < 8) u = (float)(philox4x32((uint32_t)(lane >> 2), (uint32_t)n, gid, 0xd157u, B.rs_seed_lo, B.rs_seed_hi, lane & 3) >> 8) * (1.0f / 16777216.0f);
      if (lane < 6) {
        const int kind = B.rs_dist_kind[lane];
        val = kind == 0 ? 0.0f : (kind == 1 ? B.rs_dist_range[lane][0] : B.rs_dist_range[lane][0] + (B.rs_dist_range[lane][1] - B.rs_dist_range[lane][0]) * u);
        ed[lane] = val;
      }
      const float u_int = bcast(u, 6);
      if (lane == 0) { hc[3] = 0; hc[4] = 1000 + (int)(2000.0f * u_int); hc[5] = n + 1; }
    } else if (lane == 0) hc[3] = after;
    if (lane < 6 && a.applied) gptr(const_cast<float*>(a.applied))[(size_t)env * 18 + lane] = val; /* :305 */
  }
}

/* What a wave carries from its prologue into the step: the model / batch pointers and the scalars every stage reads (in SGPRs, pinned).
 * (The per-lane model records are NOT carried: kept in registers across the re-spawn branch between load_rows and step_wave they were
 * spilled to scratch memory at their definition - step_wave fetches them in one batch as its first act.) */
struct WaveCtx {
  const GqDevModel* model; const GqDevBatch* batch;
  float h; int nlg, nfl;
#if GQ_TICKSET == 3
  long long tk[4]; /* prologue stamps (development builds) */
#endif
  int pend, lift; /* the env's pending-respawn flag (next-step auto-reset) and lift-pending flag (gq_reset's own step): fetched with the rows */
};
template <class M> __device__ __forceinline__ GqDevDofRec dof_fetch(const M& m, const int lane) {
  const int d = lane < GQ_NVD ? lane : GQ_NVD - 1;
  GqDevDofRec r;
  r.act_u = m.dof_rec[d].act_u; r.flags = m.dof_rec[d].flags; r.c_lo = m.dof_rec[d].c_lo; r.c_hi = m.dof_rec[d].c_hi;
  r.f_lo = m.dof_rec[d].f_lo; r.f_hi = m.dof_rec[d].f_hi; r.gear = m.dof_rec[d].gear; r.a_lo = m.dof_rec[d].a_lo;
  r.a_hi = m.dof_rec[d].a_hi; r.damping = m.dof_rec[d].damping; r.armature = m.dof_rec[d].armature; r.fl_row = m.dof_rec[d].fl_row;
  return r;
}
template <class M> __device__ __forceinline__ GqDevBodyRec body_fetch(const M& m, const int lane) {
  const int b = lane < GQ_NB ? lane : GQ_NB - 1;
  GqDevBodyRec r;
#pragma unroll
  for (int k = 0; k < 3; k++) r.ipos[k] = m.body_rec[b].ipos[k];
  r.mass = m.body_rec[b].mass;
#pragma unroll
  for (int k = 0; k < 6; k++) r.I[k] = m.body_rec[b].I[k];
  r.pad[0] = r.pad[1] = 0.0f;
  return r;
}
template <class M> __device__ __forceinline__ GqDevLinkRec link_fetch(const M& m, const int lane) {
  const int j = lane < GQ_NJ ? lane : GQ_NJ - 1;
  GqDevLinkRec r;
#pragma unroll
  for (int k = 0; k < 4; k++) r.bq[k] = m.link_rec[j].bq[k];
#pragma unroll
  for (int k = 0; k < 3; k++) { r.ax[k] = m.link_rec[j].ax[k]; r.jp[k] = m.link_rec[j].jp[k]; r.aloc[k] = m.link_rec[j].aloc[k]; r.r0ax[k] = m.link_rec[j].r0ax[k]; }
  r.qpos0 = m.link_rec[j].qpos0; r.pad0 = r.pad1 = r.pad2 = 0.0f;
  return r;
}

/* S0: the env's state rows and per-env scalars, global memory -> LDS, plus the solver-load hint of the env's previous step
 * (returned) and the wave's context C.  The prologue is ONE chain of three round trips, each a single batch (round 5; it was a
 * dozen: every pointer of the argument block was fetched and waited for in front of the load that used it, the pending flag was
 * waited for before the rows were requested, and the model constants of the first stages were fetched where they were used):
 *   1. the sixteen prologue pointers of the argument block: two wide scalar loads, one wait (pin);
 *   2. EVERY vector load of the prologue - pending / lift flags, state rows, scalars, the solver-load hint - plus the scalar load of
 *      the model's constants;
 *   3. the rows go to LDS.
 * Nothing loaded here stays in a register except C (an in-kernel respawn overwrites the rows and simply calls this again).
 * qfrc_applied waits in W.smooth (the actuation block adds the rest to it), the clock in W.force[0] (free until the solver).
 * user_ctrl: the caller's actions (pass 0); resets step with zero control. */
template <int SOLVER, bool PUB = false> /* PUB: the control row is a mailbox another wavefront wrote while this kernel runs (ld_pub) */
__device__ __forceinline__ int load_rows(const StepArgs& a, const StepCall& call, WaveMem& W, const int env, const bool user_ctrl, WaveCtx& C, const bool after_respawn = false) {
  const int lane = lane_id();
  /* ---- 1: pointers */
  const GqDevModel* model = a.model; const GqDevBatch* batch = a.batch;
  const double* qpos = a.qpos; const float* qvel = a.qvel; const float* warm = a.warm; const float* applied = a.applied;
  const float* time = a.time; const float* friction = a.friction; const float* cmd = a.cmd;
  const uint8_t* pending = a.pending; const uint8_t* hintp = a.load_hint; const int32_t* step_num = a.step_num; const uint8_t* liftp = a.lift_pending;
  const float* ctrl = call.ctrl;
  float hs = a.timestep; int nlg = a.nlg, nfl = a.nfl; /* (copies in the argument block: they arrive with the pointers, not behind the model pointer) */
  int auto_reset = call.auto_reset, first_pass = call.first_pass; /* (kernel arguments: pinned too, or each use re-reads the kernarg segment) */
  pin(auto_reset); pin(first_pass);
  pin(model); pin(batch); pin(qpos); pin(qvel); pin(warm); pin(applied); pin(time); pin(friction); pin(cmd);
  pin(pending); pin(hintp); pin(step_num); pin(liftp); pin(ctrl);
  pin(hs); pin(nlg); pin(nfl);
#if GQ_TICKSET == 3
  C.tk[0] = cycles();
#endif
  /* ---- 2: one batch of loads */
  int pend = 0, lift = 0;
  if (auto_reset == 2 && pending) pend = (int)ldv<PUB>(pending + env);
  if (first_pass && liftp) lift = (int)ldv<PUB>(liftp + env);
  const int l19 = lane < 19 ? lane : 18, l18 = lane < 18 ? lane : 17, l12 = lane < 12 ? lane : 11, l4 = lane < 4 ? lane : 3;
  const double q = ldv<PUB>(qpos + (size_t)env * 19 + l19);
  const float qv = ldv<PUB>(qvel + (size_t)env * 18 + l18), wm = ldv<PUB>(warm + (size_t)env * 18 + l18);
  float ap = 0.0f, ct = 0.0f, cm = 0.0f, mu = -1.0f;
  if (applied) ap = ldv<PUB>(applied + (size_t)env * 18 + l18);
  if (ctrl && user_ctrl) {
    if constexpr (PUB) ct = ld_pub(ctrl + (size_t)env * 12 + l12);
    else ct = gptr(ctrl)[(size_t)env * 12 + l12];
  }
  if (cmd) cm = ldv<PUB>(cmd + (size_t)env * 4 + l4);
  if (friction) mu = ldv<PUB>(friction + env);
  const int sn = ldv<PUB>(step_num + env);
  const float tm = ldv<PUB>(time + env);
  int hint = 0;
  if (SOLVER == 1 && hintp) hint = (int)ldv<PUB>(hintp + env);
#if GQ_TICKSET == 3
  C.tk[1] = cycles();
#endif
  C.model = model; C.batch = batch; C.h = hs; C.nlg = nlg; C.nfl = nfl;
  /* ---- 3: rows -> LDS */
  if (lane < 19) {
    if (lane < 2) W.bxy[lane] = q;
    else if (lane == 2) W.basez = (float)q;
    else if (lane < 7) W.qb[lane - 3] = (float)q;
    else W.qj[lane - 7] = (float)q;
  }
  if (lane < 18) { W.qvel[lane] = qv; W.warm[lane] = wm; W.smooth[lane] = ap; }
  if (lane < 12) W.ctrl[lane] = ct;
  if (lane < 4) W.cmd[lane] = cm;
  if (lane == 0) { W.mu_env = mu; W.step_old = sn; W.force[0] = tm; }
  /* whether the reset's lift loop is due in this step's S6 waits in LDS (carried in a register across the re-spawn branch it was spilled to
   * scratch memory): from the reset kernel's flag here, from reset_wave itself after an in-kernel re-spawn */
  if (!after_respawn && lane == 0) GQ_LIFT_DUE(W) = first_pass ? lift : 0;
  C.pend = uniform(pend); C.lift = uniform(lift);
  hint = uniform(hint);
#if GQ_TICKSET == 3
  __builtin_amdgcn_s_waitcnt(0);
  C.tk[2] = cycles();
#endif
  return hint;
}

/* One mj_step + observation epilogue for this wave's env.  pass 0: the user's step.  pass 1: the reset's own step
 * (zero control, friction committed afterwards, termination flags of pass 0 are kept).  pass 2: the reset's own step of
 * a next-step auto-reset (as pass 1, flags cleared).  Returns `terminated`. */
/* SOLVER 0: PGS (mj_solPGS), 1: Newton (mj_solNewton, MuJoCo's default).  MODE 0: production; 2: production + the
 * GQ_STOP_STAGE cut; 1 (DBG): the variant with the debug record, the
 * stage timers and the GQ_STOP_STAGE cut compiled in - the production variant carries none of it (no timer
 * accumulators or row data kept live for the record: they cost registers inside the solver loop).
 * CONE: elliptic friction cones (Newton only): contacts take dim rows [n, t1, t2, torsion, roll1, roll2].
 * BOXES: the scene has static world boxes (gq_boxes.h; Newton only): contacts carry their own normal.
 * SELF: robot self-collision (Newton only): contacts between two bodies of the robot, general frames, two-body rows.
 * PRIM (BOXES variants): the robot has sphere / capsule / box link geoms, whose contacts with world boxes and with each other
 * come from the exact pair routines (gq_pairs.h); robots of hulls only get the variant without that code - merely compiled
 * in, it cost them 17 % (registers spilled across the box loop). */
template <int SOLVER, int MODE, bool CONE, bool BOXES, bool SELF, bool PRIM, bool PUB = false> /* PUB: the observation row is published to a
                                                                                                   * concurrently running reader (st_pub) */
__device__ __forceinline__ int step_wave(const StepArgs& a, const StepCall& call, WaveMem& W, const int pass, const int hint, const WaveCtx& C, const long long t_entry = 0) {
  /* lane / env are made opaque so that per-lane address arithmetic is not hoisted out of the (rarely taken) second
   * pass loop of the kernel and kept live - that hoisting alone cost > 250 spilled VGPRs */
  int lane_o = lane_id(), env_o = wave_index() + uniform(call.env0);
  opaque(lane_o); opaque_s(env_o);
  const int lane = lane_o, env = env_o;
  constexpr bool DBG = MODE == 1;
  constexpr bool GEN = BOXES || SELF; /* contacts carry their own frame and may join two bodies of the robot */
  const GQ_MODEL GqDevModel& m = *mptr(C.model);
  const GQ_MODEL GqDevBatch& Bt = *mptr(C.batch);
  const float h = C.h;
  /* the record describes the forward pass whose results the caller sees: the user's step, the reset's own step of
   * gq_reset, or the reset step of a next-step auto-reset - not the second pass of a same-step auto-reset */
  const bool rec_pass = pass == call.first_pass || pass == 2;
  const bool timing = DBG && call.debug && rec_pass && env < Bt.debug_envs;
  const long long t_start = timing ? ((GQ_TICKSET && t_entry) ? t_entry : cycles()) : 0; /* sub-stage builds count from kernel entry */
#if GQ_TICKSET
  if (lane == 0) { W.tk_T = timing ? call.debug + (size_t)env * GQ_DBG_SIZE + GQ_DBG_TIMER : nullptr; W.tk_t0 = t_start; }
#endif
#define GQ_TICK(i) do { if constexpr (DBG && (GQ_TICKSET == 0 || (i) == 13 || (i) == 15)) { \
    if (timing && lane == 0) call.debug[(size_t)env * GQ_DBG_SIZE + GQ_DBG_TIMER + (i)] = (float)(cycles() - t_start); } \
    if constexpr (MODE == 2) { if (call.stop_stage == (i) && pass != 1) return 0; } } while (0)

  if constexpr (DBG) if (timing && lane == 0) { /* where and when this wave runs: slots 24-27 of the time stamps */
    float* T = call.debug + (size_t)env * GQ_DBG_SIZE + GQ_DBG_TIMER;
    T[24] = (float)(wall_clock64() & 0xFFFFF);                                  /* 100 MHz clock common to the device */
    T[26] = (float)(__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0xFFFF);        /* HW_ID: wave, simd, pipe, cu, sh, se */
    T[27] = (float)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xF);          /* XCC_ID */
  }
  GQ_TICK(15); /* marker 15: nothing done yet - the launch floor */
#if GQ_TICKSET == 3
  if (timing && lane == 0) { float* T_ = call.debug + (size_t)env * GQ_DBG_SIZE + GQ_DBG_TIMER; T_[1] = (float)(C.tk[0] - t_start); T_[2] = (float)(C.tk[1] - t_start); T_[3] = (float)(C.tk[2] - t_start); T_[4] = (float)(cycles() - t_start); }
#endif
  /* the per-lane model records of the first three stages, one batch of wide loads (lane = dof: actuation, damping, armature; lane = link:
   * local transform; lane = body: inertia): they were nine + five + three dependent memory round trips in front of their arithmetic */
  const GqDevDofRec Drec = dof_fetch(m, lane);
  const GqDevLinkRec Lrec = link_fetch(m, lane);
  const GqDevBodyRec Brec = body_fetch(m, lane);
  /* ================================================================ S0: the env's rows wait in LDS (load_rows) */
  /* scheduling hint: an env whose previous step needed several Newton iterations will most likely need them again; its
   * wave gets issue priority from the start (the launch lasts as long as its slowest wave) */
  const int prio_hint = pass != 0 ? 3 : hint;
  bool fwd_only = false; /* gq_forward (mj_step1 / mj_forward): no state is advanced */
  if constexpr (DBG) fwd_only = call.forward != 0;
  wave_barrier();
  GQ_SUB(W, 1, 0); /* kernel entry -> rows in LDS */
  GQ_SUB(W, 3, 4);
  /* the clock and the step counter advance here (stores only: their old values came with the rows) */
  {
    int32_t* s_num = a.step_num; int32_t* s_prev = a.step_prev; float* s_time = a.time;
    pin(s_num); pin(s_prev); pin(s_time);
    if (!fwd_only) { /* (every lane stores the same word: no exec-mask block - see the note on mirror lanes below) */
      const int32_t sn = W.step_old;
      gptr(s_num)[env] = sn + 1;
      if (s_prev) gptr(s_prev)[env] = sn;
      gptr(s_time)[env] = W.force[0] + h;
    }
  }
  /* MIRROR LANES (round 5).  A block `if (lane < N) { ... LDS[lane] = f(LDS[lane]) }` costs a lone wavefront ~50 cycles of exec-mask
   * bookkeeping (v_cmp, s_and_saveexec, s_cbranch_execz, s_or: profiles/r05_issue_model_lone_wave.txt) around a handful of 4-cycle
   * instructions, and a step had ~200 of them.  Where a block has no cross-lane primitive inside, it now runs UNCONDITIONALLY with
   * the lane index clamped: lanes >= N repeat lane N - 1's reads, arithmetic and stores - same addresses, same values, which the LDS and
   * the memory pipeline take at the cost of one lane's (measured).  A block that reads and rewrites the same word puts a wave_barrier()
   * between the read and the write (free on the GPU; the host emulator runs its lanes one after the other between barriers). */
  const int ld = lane < GQ_NVD ? lane : GQ_NVD - 1, lj = lane < GQ_NJ ? lane : GQ_NJ - 1, lb = lane < GQ_NB ? lane : GQ_NB - 1, lq = lane < 4 ? lane : 3;

  /* actuation (mj_fwdActuation: torque motors) and passive damping depend on ctrl / qvel and model constants only: done
   * here, so that the (two-level dependent) model loads overlap with the kinematics instead of sitting on S5's path */
  { /* lane = dof (mirror lanes) - the dof's record came in one batch; the limits are selects, not branches */
    const GqDevDofRec& D = Drec;
    float act = 0.0f;
    {
      const int u = D.act_u;
      float c = W.ctrl[u >= 0 ? u : 0];
      c = (D.flags & 1) ? fminf(fmaxf(c, D.c_lo), D.c_hi) : c;
      c = (D.flags & 2) ? fminf(fmaxf(c, D.f_lo), D.f_hi) : c;
      act = u >= 0 ? D.gear * c : 0.0f;
      act = (D.flags & 4) ? fminf(fmaxf(act, D.a_lo), D.a_hi) : act;
    }
    W.act[ld] = act;
    const float damp = D.damping;
    if constexpr (SOLVER == 1) W.F[0][ld] = h * damp; /* the Newton path stores no factor: F keeps h*damping for the Euler system (S10) */
    const float sm_new = -damp * W.qvel[ld] + act + W.smooth[ld]; /* qfrc_applied waits there */
    wave_barrier();
    W.smooth[ld] = sm_new;
  }
  GQ_SUB(W, 1, 1); /* actuation + passive */
  if constexpr (SOLVER == 1) wave_priority(prio_hint);
  if constexpr (DBG) if (timing && lane == 0) call.debug[(size_t)env * GQ_DBG_SIZE + GQ_DBG_TIMER + 28] = (float)prio_hint;
  stage_kinematics(W, Lrec);

  GQ_TICK(1); GQ_SUB(W, 1, 3);
  /* ================================================================ S2: spatial inertias about O = base origin */
  int s3e[3]; float s3a[3]; /* the lane's three entries of S3: fetched now, in flight during S2 */
#pragma unroll
  for (int p = 0; p < 3; p++) { s3e[p] = m.s3_ent[p][lane]; s3a[p] = m.s3_arm[p][lane]; }
  const V3 O = v3(0.0f, 0.0f, W.basez);
  { /* lane = body, mirror lanes */
    const int b = lb;
    const float* R = W.xmat[b];
    V3 d = ld3(W.xpos[b]) + matvec(R, ld3(Brec.ipos)) - O;
    /* I_w = R Ib R' */
    const float* Ib = Brec.I;
    float A[9] = {Ib[0], Ib[3], Ib[4], Ib[3], Ib[1], Ib[5], Ib[4], Ib[5], Ib[2]}, T[9], Iw[9];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = 0; c < 3; c++) T[3 * r + c] = R[3 * r] * A[c] + R[3 * r + 1] * A[3 + c] + R[3 * r + 2] * A[6 + c];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = 0; c < 3; c++) Iw[3 * r + c] = T[3 * r] * R[3 * c] + T[3 * r + 1] * R[3 * c + 1] + T[3 * r + 2] * R[3 * c + 2];
    float mb = Brec.mass, dd = dot(d, d);
    float* ci = W.u.dyn.cinert[b];
    ci[0] = Iw[0] + mb * (dd - d.x * d.x); ci[1] = Iw[4] + mb * (dd - d.y * d.y); ci[2] = Iw[8] + mb * (dd - d.z * d.z);
    ci[3] = Iw[1] - mb * d.x * d.y; ci[4] = Iw[2] - mb * d.x * d.z; ci[5] = Iw[5] - mb * d.y * d.z;
    ci[6] = mb * d.x; ci[7] = mb * d.y; ci[8] = mb * d.z; ci[9] = mb;
  }
  /* motion subspaces: lanes 0-5 the base dofs (translations e_k; body-fixed rotation axes through O, whose linear part vanishes), selects */
  {
    const int d6 = lane < 6 ? lane : 5, c3 = d6 < 3 ? 0 : d6 - 3;
    const float* R = W.xmat[0];
    const float r0 = R[c3], r1 = R[3 + c3], r2 = R[6 + c3];
    const bool tr = d6 < 3;
    float* s = W.cdof[d6];
    s[0] = tr ? 0.0f : r0; s[1] = tr ? 0.0f : r1; s[2] = tr ? 0.0f : r2;
    s[3] = d6 == 0 ? 1.0f : 0.0f; s[4] = d6 == 1 ? 1.0f : 0.0f; s[5] = d6 == 2 ? 1.0f : 0.0f;
  }
  wave_barrier();
  { /* lane - 6 = hinge, mirror lanes on both sides */
    const int j = lane < 6 ? 0 : (lane < GQ_NVD ? lane - 6 : GQ_NJ - 1);
    V3 ax = ld3(W.u.dyn.axis[j]);
    st3(W.cdof[6 + j], ax);
    st3(W.cdof[6 + j] + 3, cross(ax, O - ld3(W.u.dyn.anchor[j])));
  }
  /* composite inertias: everything is about the same point in the same axes, so they are plain sums */
  {
    const int l40 = lane < 40 ? lane : 39;
    const int leg = l40 / 10, k = l40 % 10, b0 = 1 + 3 * leg;
    float c2 = W.u.dyn.cinert[b0 + 2][k], c1 = W.u.dyn.cinert[b0 + 1][k] + c2, c0 = W.u.dyn.cinert[b0][k] + c1;
    W.u.dyn.crb[b0 + 2][k] = c2; W.u.dyn.crb[b0 + 1][k] = c1; W.u.dyn.crb[b0][k] = c0;
  }
  wave_barrier();
  { const int l10 = lane < 10 ? lane : 9; W.u.dyn.crb[0][l10] = W.u.dyn.cinert[0][l10] + W.u.dyn.crb[1][l10] + W.u.dyn.crb[4][l10] + W.u.dyn.crb[7][l10] + W.u.dyn.crb[10][l10]; }
  wave_barrier();

  GQ_TICK(2); GQ_SUB(W, 1, 4);
  /* ================================================================ S3: joint-space inertia */
  /* lane = stored entry (three per lane: 108 of the legs' rows + the full 6x6 base block), all of them independent - every lane's LDS
   * reads are in flight together.  (Round 4: lane = dof walked up its ancestors in a divergent loop, one LDS round trip and two
   * exec-mask branches per ancestor: 5 k cycles for a lone wave.)  Same operands in the same order: the values are the old ones. */
  {
    static_assert(sizeof(WaveMem::Mc) == 108 * 4 && offsetof(WaveMem, Mb) == offsetof(WaveMem, Mc) + sizeof(WaveMem::Mc), "S3 writes Mc | Mb as one flat array");
    float* M0 = &W.Mc[0][0];
#pragma unroll
    for (int p = 0; p < 3; p++) { /* (the table's slots past entry 143 repeat entry 143: mirror lanes) */
      const int ent = s3e[p], dd = ent & 0xff, sa = (ent >> 8) & 0xff, bd = (ent >> 16) & 0xff;
      float buf[6];
      mul_inert(buf, W.u.dyn.crb[bd], W.cdof[dd]);
      const float* sv = W.cdof[sa];
      const float v = sv[0] * buf[0] + sv[1] * buf[1] + sv[2] * buf[2] + sv[3] * buf[3] + sv[4] * buf[4] + sv[5] * buf[5];
      const int e = lane + GQ_WAVE * p;
      M0[e < 143 ? e : 143] = (ent >> 24) ? v + s3a[p] : 0.0f;
    }
  }
  wave_barrier();

  GQ_TICK(3); GQ_SUB(W, 1, 5);
  /* ================================================================ S4: factorise M and M + h*D */
  /* (the Newton path solves its three systems with the fused elimination (gq_newton.h) and stores no factor.  PGS on a scene with world
   * geoms / robot self-collision: the collision stages use the factors' LDS as scratch (gq_boxes.h GQ_BX_*), so the factors are taken
   * after the rows are built - see S8) */
  if constexpr (SOLVER == 0 && !(BOXES || SELF)) factor_tree_both(W, m.dof_damping, h);

  GQ_TICK(4);
  /* the lane's collision item (S6: lane = item) is fetched now: its loads are in flight during the velocity stage */
  const int nlg = C.nlg;
  const ItemRegs IT = item_fetch(m, lane < 4 + nlg ? lane : 0);
  /* with it, everything else S6 - S9 read from the model per lane (joint-limit record, friction-loss row, the lane's two Hessian entries,
   * the foot record) and per wave (StepConsts: scalar loads, pinned) - one batch in front of the velocity stage */
  const FootRec FRec = foot_fetch(m, lane);
  const int l12_ = lane < GQ_NJ ? lane : GQ_NJ - 1, l18_ = lane < GQ_NVD ? lane : GQ_NVD - 1;
  /* (elliptic variants fetch these where they are used: their solver needs the registers, and what is prefetched here was spilled to scratch
   * memory on the way - 92 bytes per lane after the early-fetch rework of round 5, 28 before it) */
  constexpr bool EARLY = !CONE && !(BOXES && PRIM); /* (the world-geom variants of the primitive-geom robots: 44 bytes of scratch with the early fetch, see below) */
#define GQ_FETCH_LIM() do { lim_on = m.lim_rec[l12_].limited; lim_lo_ = m.lim_rec[l12_].lo; lim_hi_ = m.lim_rec[l12_].hi; lim_mg = m.lim_rec[l12_].margin; } while (0)
#define GQ_FETCH_FLR() do { flr_dof = m.fl_row[l18_].dof; flr_R = m.fl_row[l18_].R; flr_B = m.fl_row[l18_].B; flr_floss = m.fl_row[l18_].floss; } while (0)
#define GQ_FETCH_HENT() do { if constexpr (SOLVER == 1) { hent_pre[0] = m.newton_hent[0][lane]; hent_pre[1] = m.newton_hent[1][lane]; } else { hent_pre[0] = hent_pre[1] = 0; } } while (0)
  int lim_on = 0, flr_dof = 0, hent_pre[2] = {0, 0};
  float lim_lo_ = 0.0f, lim_hi_ = 0.0f, lim_mg = 0.0f, flr_R = 0.0f, flr_B = 0.0f, flr_floss = 0.0f;
  if constexpr (EARLY) { GQ_FETCH_LIM(); GQ_FETCH_FLR(); GQ_FETCH_HENT(); }
  /* (the late fetches go through closures: the same statements written in line left the elliptic variants with 60 instead of 48 bytes of scratch -
   * the register allocator's outcome on these kernels is that sensitive; the pyramidal variants keep the in-line form they were tuned with) */
  auto fetch_lim = [&]() { GQ_FETCH_LIM(); };
  auto fetch_flr = [&]() { GQ_FETCH_FLR(); };
  auto fetch_hent = [&]() { GQ_FETCH_HENT(); };
  StepConsts K;
  {
    int fl0 = m.hot_foot_leg[0], fl1 = m.hot_foot_leg[1], fl2 = m.hot_foot_leg[2], fl3 = m.hot_foot_leg[3], its = m.iterations, nsp = m.hot_nsp, scut = m.hot_self_cut;
    float fmu = m.hot_floor_mu, imr = m.impratio, gz = m.gravity_z, mi = m.meaninertia, tol = m.tolerance, nf = m.noise_floor, smg = m.hot_self_margin;
    const GQ_MODEL float* vxq = mptr(a.vx); const GQ_MODEL float* vyq = mptr(a.vy); const GQ_MODEL float* vzq = mptr(a.vz);
    pin(fl0); pin(fl1); pin(fl2); pin(fl3); pin(its); pin(nsp); pin(scut); pin(fmu); pin(imr); pin(gz); pin(mi); pin(tol); pin(nf); pin(smg);
    pin(vxq); pin(vyq); pin(vzq);
    K.foot_leg[0] = fl0; K.foot_leg[1] = fl1; K.foot_leg[2] = fl2; K.foot_leg[3] = fl3; K.iterations = its; K.nsp = nsp; K.self_cut = scut;
    K.floor_mu = fmu; K.impratio_rs = fast_rsqrt(imr); K.gravity_z = gz; K.nw_scale = fast_rcp(mi * 18.0f); K.tolerance = tol; K.noise_floor = nf; K.self_margin = smg;
    K.vx = vxq; K.vy = vyq; K.vz = vzq;
  }
  const GQ_MODEL float* vx_p = K.vx; const GQ_MODEL float* vy_p = K.vy; const GQ_MODEL float* vz_p = K.vz;
  /* ================================================================ S5: velocity stage (mj_comVel, mj_rne) */
  { /* lane = leg, mirror lanes */
    /* base velocity and bias acceleration, recomputed per leg lane */
    float vb[6], ab[6];
    {
      const float* R = W.xmat[0];
      V3 wl = v3(W.qvel[3], W.qvel[4], W.qvel[5]);
      V3 ww = matvec(R, wl), vl = v3(W.qvel[0], W.qvel[1], W.qvel[2]);
      st3(vb, ww); st3(vb + 3, vl);
      V3 al = cross(vl, ww);
      ab[0] = ab[1] = ab[2] = 0.0f; ab[3] = al.x; ab[4] = al.y; ab[5] = al.z - K.gravity_z;
#pragma unroll
      for (int k = 0; k < 6; k++) { W.u.dyn.cvel[0][k] = vb[k]; W.u.dyn.cacc[0][k] = ab[k]; } /* (every lane: same words, same values) */
    }
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const int b = 1 + 3 * lq + i, d = 6 + 3 * lq + i;
      float cd[6];
      cross_motion(cd, vb, W.cdof[d]);
      float qd = W.qvel[d];
#pragma unroll
      for (int k = 0; k < 6; k++) { vb[k] += W.cdof[d][k] * qd; ab[k] += cd[k] * qd; W.u.dyn.cvel[b][k] = vb[k]; W.u.dyn.cacc[b][k] = ab[k]; }
    }
  }
  wave_barrier();
  { /* lane = body, mirror lanes */
    float t1[6], t2[6], f[6];
    mul_inert(t1, W.u.dyn.cinert[lb], W.u.dyn.cacc[lb]);
    mul_inert(t2, W.u.dyn.cinert[lb], W.u.dyn.cvel[lb]);
    cross_force(f, W.u.dyn.cvel[lb], t2);
#pragma unroll
    for (int k = 0; k < 6; k++) W.u.dyn.cfrc[lb][k] = f[k] + t1[k];
  }
  wave_barrier();
  { /* accumulate up the legs: lane = (leg, component), mirror lanes; the sums are read first, written after a barrier */
    const int l24 = lane < 24 ? lane : 23;
    const int leg = l24 / 6, k = l24 % 6, b0 = 1 + 3 * leg;
    float f2 = W.u.dyn.cfrc[b0 + 2][k], f1 = W.u.dyn.cfrc[b0 + 1][k] + f2, f0 = W.u.dyn.cfrc[b0][k] + f1;
    wave_barrier();
    W.u.dyn.cfrc[b0 + 1][k] = f1; W.u.dyn.cfrc[b0][k] = f0;
  }
  wave_barrier();
  {
    const int l6 = lane < 6 ? lane : 5;
    const float fb = W.u.dyn.cfrc[0][l6] + (W.u.dyn.cfrc[1][l6] + W.u.dyn.cfrc[4][l6] + W.u.dyn.cfrc[7][l6] + W.u.dyn.cfrc[10][l6]);
    wave_barrier();
    W.u.dyn.cfrc[0][l6] = fb;
  }
  wave_barrier();
  { /* lane = dof, mirror lanes */
    const float* s = W.cdof[ld];
    const float* f = W.u.dyn.cfrc[dof_body(ld)];
    float bias = s[0] * f[0] + s[1] * f[1] + s[2] * f[2] + s[3] * f[3] + s[4] * f[4] + s[5] * f[5];
    const float sm_new = W.smooth[ld] - bias; /* passive + actuation + applied were put there right after S0 */
    wave_barrier();
    W.bias[ld] = bias;
    W.smooth[ld] = sm_new;
  }

  GQ_TICK(5); GQ_SUB(W, 1, 6);
  /* ================================================================ S6: collision with the floor (z = 0) */
  SelfPrefetch self_pre;
  if constexpr (SELF && !BOXES) self_pre = self_prefetch(m, nlg, K.nsp); /* (world-box variants fetch it behind the box loop: 18 registers less across it) */
  stage_collision_scan(W, m, vx_p, vy_p, vz_p, false, FRec);
  GQ_SUB(W, 1, 7); /* hull cloud scan */
  /* reset on a scene without world boxes / height field: the lift loop of QuadrupedEnv.reset (quadruped_env.py:376-388:
   * z += 1.1 max|dist| until no foot-body contact, <= 100 iterations) runs HERE, on the distances this step's own
   * kinematics and collision scan just produced, instead of on a kinematics + scan pass of its own inside reset_wave: on
   * the flat floor a lift by dz moves every distance by dz and leaves everything that S2-S5 computed (all of it relative
   * to the base origin) untouched, so the reset's mj_step simply continues from the lifted pose */
  /* lane = collision item in MuJoCo's order (increasing geom id; con_order interleaves feet and link geoms): its candidate
   * contact points with the floor, up to four (floor_candidates) */
  const int nitem = 4 + nlg;
  FloorCand FC;
  FC.n = 0; FC.r = 0.0f; FC.t1c = 0.0f; FC.t1s = 1.0f; FC.key = 0xE4;
#pragma unroll
  for (int k = 0; k < 4; k++) { FC.dist[k] = 1e30f; FC.pt[k] = v3(0.0f, 0.0f, 0.0f); }
  int code = 0, body = 0, dim = 3, fric_rule = 0;
  bool calf = false;
  float cmargin = 0.0f, inc = 0.0f, fgeom = 0.0f;
  float solref[2] = {0.02f, 1.0f}, solimp[5] = {0.9f, 0.95f, 0.001f, 0.5f, 2.0f};
  if (lane < nitem) {
    floor_candidates(W, IT, FC);
    code = IT.code; cmargin = IT.margin; body = IT.body; dim = IT.dim; inc = IT.inc; calf = IT.calf != 0;
    fgeom = IT.friction0; fric_rule = IT.fric_rule;
    solref[0] = IT.solref[0]; solref[1] = IT.solref[1];
#pragma unroll
    for (int q = 0; q < 5; q++) solimp[q] = IT.solimp[q];
  }
  if (uniform(GQ_LIFT_DUE(W))) { /* wave-uniform; only set on scenes without world boxes / height field */
    /* the reference lifts by 1.1 max |contact.dist| over EVERY contact of the calf bodies with the ground (feet_contact_state
     * lists them all, quadruped_env.py:378-385) until none is left */
    float dz = 0.0f;
    for (int it = 0; it < 100; it++) {
      float pen = 0.0f;
      bool touching = false;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const bool tk = calf && k < FC.n && FC.dist[k] + dz < cmargin;
        touching = touching || tk;
        pen = tk ? fmaxf(pen, fabsf(FC.dist[k] + dz)) : pen;
      }
      if (ballot(touching) == 0) break;
      dz += 1.1f * wave_max(pen);
    }
    bool still = false;
#pragma unroll
    for (int k = 0; k < 4; k++) still = still || (calf && k < FC.n && FC.dist[k] + dz < cmargin);
    const int failed = ballot(still) != 0;
    wave_barrier();
    if (lane == 0) { W.basez += dz; if (a.lift_failed) gptr(a.lift_failed)[env] = (uint8_t)failed; }
    if (lane < GQ_NB) W.xpos[lane][2] += dz;
    if (lane < 4) W.foot_world[lane][2] += dz;
#pragma unroll
    for (int k = 0; k < 4; k++) { FC.dist[k] += dz; FC.pt[k].z += dz; }
    wave_barrier();
  }
  GQ_TICK(14); GQ_SUB(W, 1, 8); /* floor candidates (+ lift) */
  /* the dynamics row (gq_batch_set_outputs): what the reference reads from mjData after the step for model-based control -
   * mj_fullM, qfrc_bias, body poses, the foot points - straight out of LDS, production kernel */
  if (a.dyn && rec_pass) { /* wave-uniform */
    GQ_GLOBAL float* D = gptr(a.dyn) + (size_t)env * GQ_DYN_STRIDE;
    const float* Mc0 = &W.Mc[0][0];
    const float* Mb0 = &W.Mb[0][0];
    const float* xp0 = &W.xpos[0][0];
    const float* xm0 = &W.xmat[0][0];
    D[GQ_DYN_MC + lane] = Mc0[lane];
    if (lane < 108 - GQ_WAVE) D[GQ_DYN_MC + GQ_WAVE + lane] = Mc0[GQ_WAVE + lane];
    if (lane < 36) D[GQ_DYN_MB + lane] = Mb0[lane];
    if (lane < GQ_NVD) D[GQ_DYN_BIAS + lane] = W.bias[lane];
    if (lane < 39) D[GQ_DYN_XPOS + lane] = xp0[lane];
    D[GQ_DYN_XMAT + lane] = xm0[lane];
    if (lane < 117 - GQ_WAVE) D[GQ_DYN_XMAT + GQ_WAVE + lane] = xm0[GQ_WAVE + lane];
    if (lane < 12) D[GQ_DYN_FOOT + lane] = (&W.foot_world[0][0])[lane];
  }
  /* contact list in MuJoCo's order, capped.  Lane `it` owns the (up to four) contacts of collision item `it`; ranks and row
   * offsets come from one wave prefix sum over (contacts, rows, reserved virtual rows) packed into an int - no serial section. */
  {
    const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    bool tk[4];
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { tk[k] = lane < nitem && k < FC.n && FC.dist[k] < cmargin; cnt += tk[k] ? 1 : 0; }
    const bool touching = cnt > 0;
    /* _check_for_invalid_contacts (quadruped_env.py:1228-1248): body-level test, before any capping */
    const int invalid = ballot(touching && !calf) != 0;
    int ftm = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) ftm |= (ballot(touching && body == 3 + 3 * K.foot_leg[k]) != 0) ? (1 << k) : 0;
    /* joint limits: lane j < 12 owns hinge j (lower side first, then upper) */
    bool lim_lo = false, lim_hi = false;
    float dlo = 0.0f, dhi = 0.0f;
    if constexpr (!EARLY) fetch_lim();
    {
      const float q = W.qj[lj];
      dlo = q - lim_lo_; dhi = lim_hi_ - q;
      const bool mine = lane < GQ_NJ && lim_on != 0;
      lim_lo = mine && dlo < lim_mg; lim_hi = mine && dhi < lim_mg;
    }
    const uint64_t mlo = ballot(lim_lo), mhi = ballot(lim_hi);
    int nl = popc64(mlo) + popc64(mhi);
    if (nl > 0) { /* wave-uniform */
      int at = popc64(mlo & lt) + popc64(mhi & lt);
      if (lim_lo && at < GQ_NJ) { W.u2.c.lim_jnt[at] = lane; W.u2.c.lim_side[at] = 1.0f; W.u2.c.lim_dist[at] = dlo; at++; }
      if (lim_hi && at < GQ_NJ) { W.u2.c.lim_jnt[at] = lane; W.u2.c.lim_side[at] = -1.0f; W.u2.c.lim_dist[at] = dhi; }
      if (nl > GQ_NJ) nl = GQ_NJ;
    }
    /* row budget: friction rows, limit rows, then whole contacts in order while they fit (a prefix of the list).
     * pyramidal: 2 (dim - 1) edge rows; elliptic: dim rows + (dim - 1) rows of LDS above nefc reserved per cone contact for
     * the virtual rows of its Hessian block (gq_newton.h) */
    const int need = dim == 1 ? 1 : (CONE ? dim : 2 * (dim - 1));
    const int vres = (CONE && need > 1) ? need - 1 : 0;
    /* <= 42 items x 4 contacts x 6 rows: the three running sums stay inside their bit fields (8 / 10 / 10 bits) */
    const int packed = cnt | ((cnt * need) << 8) | ((cnt * vres) << 18);
    const int incl = wave_incl_scan(packed);
    const int excl = incl - packed;
    const int idx0 = excl & 0xff, rows0 = C.nfl + nl + ((excl >> 8) & 0x3ff), res0 = (excl >> 18) & 0x3ff;
    const float mu_env = W.mu_env;
    const float ff = mu_env >= 0.0f ? mu_env : K.floor_mu;
    /* friction mixing; _set_ground_friction overrides floor and feet with [mu, 0.005, 0] (quadruped_env.py:1292) */
    const float fg = (code < 4 && mu_env >= 0.0f) ? mu_env : fgeom;
    const float mu = fmaxf(1e-5f, fric_rule == 0 ? fmaxf(ff, fg) : (fric_rule == 1 ? ff : fg)); /* mjMINMU */
    int nfit = 0;
    const bool multi = ballot(tk[1] || tk[2] || tk[3]) != 0; /* wave-uniform: some item touches at a point other than its first (primitive geoms only) */
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (k > 0 && !multi) break;
      if (tk[k]) { /* the lane's j-th contact in MuJoCo's order: j = touching candidates placed before this one */
        int j = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) j += (q != k && tk[q] && ((FC.key >> (2 * q)) & 3) < ((FC.key >> (2 * k)) & 3)) ? 1 : 0;
        const int idx = idx0 + j, row0 = rows0 + j * need, res = res0 + (j + 1) * vres;
        const bool fits = idx < GQ_MAXCON && row0 + need + res <= (CONE ? 64 : GQ_MAXEFC) && row0 + need <= GQ_MAXEFC;
        if (fits) {
          const float dist = FC.dist[k];
          W.con_geom[idx] = code; W.con_body[idx] = body; W.con_dim[idx] = dim; W.con_row[idx] = row0;
          W.con_dist[idx] = dist; W.con_inc[idx] = inc; W.con_mu[idx] = mu;
          W.con_pos[idx][0] = FC.pt[k].x; W.con_pos[idx][1] = FC.pt[k].y; W.con_pos[idx][2] = FC.pt[k].z - (FC.r + 0.5f * dist);
          W.con_solref[idx][0] = solref[0]; W.con_solref[idx][1] = solref[1];
#pragma unroll
          for (int q = 0; q < 5; q++) W.con_solimp[idx][q] = solimp[q];
          W.con_t1[idx][0] = FC.t1c; W.con_t1[idx][1] = FC.t1s;
          nfit++;
        }
      }
    }
    const int tot = bcast(wave_incl_scan(nfit | ((nfit * need) << 8)), 63);
    { /* (every lane stores the same words) */
      W.ndrop = (bcast(incl, 63) & 0xff) - (tot & 0xff); /* lane 63's inclusive sum = every touching candidate of the floor pass */
      W.ncon = tot & 0xff; W.nlim = nl; W.nefc = C.nfl + nl + (tot >> 8); W.invalid = invalid;
      W.foot_touch = ftm;
    }
  }
  wave_barrier();
  GQ_SUB(W, 1, 9); /* floor contact list, limits */
  if constexpr (BOXES) {
    const double bx0 = W.bxy[0], by0 = W.bxy[1]; /* base x/y of this forward pass, f64 */
    const float mu_b = W.mu_env;
    stage_box_contacts<CONE, SELF, PRIM>(W, m, vx_p, vy_p, vz_p, bx0, by0, mu_b, self_pre, IT, K, nlg, Bt, (DBG && timing) ? call.debug + (size_t)env * GQ_DBG_SIZE + GQ_DBG_XQ : nullptr, env);
  } else if constexpr (SELF) {
    const float mu_b = W.mu_env;
    stage_self_contacts<CONE>(W, m, mu_b, self_pre, K, nlg, Bt, (DBG && timing) ? call.debug + (size_t)env * GQ_DBG_SIZE + GQ_DBG_XQ : nullptr, env);
  }
  const int nefc = uniform(W.nefc), ncon = uniform(W.ncon), nlim = uniform(W.nlim), nfl = C.nfl; /* SGPRs */
  if (timing) { /* body poses go to the debug record now: xmat's LDS is reused by the Newton solver */
    float* D = call.debug + (size_t)env * GQ_DBG_SIZE;
    if (lane < 39) D[GQ_DBG_XPOS + lane] = W.xpos[lane / 3][lane % 3];
    for (int k = lane; k < 117; k += GQ_WAVE) D[GQ_DBG_XMAT + k] = W.xmat[k / 9][k % 9];
  }

  GQ_TICK(6); GQ_SUB(W, 1, 12); GQ_SUB(W, 2, 0);
  /* ================================================================ S7: constraint rows, lane = row */
  /* per-lane row descriptor first (what kind of row, which dof / contact direction), then ONE unrolled sweep over the
   * 18 dofs produces the J entries for every row kind at once: J[k] = cdof[k] . [p x dir ; dir] on the contact's chain,
   * +-1 at the dof of a friction-loss / limit row.  The Newton path streams the entries straight into LDS (its solver
   * only reads J from there) - no 18-register row is kept live across the row set-up; PGS keeps them in registers. */
  int rtype = ROW_NONE;
  float rpos = 0.0f, rmargin = 0.0f, rfloss = 0.0f, rdiag = 0.0f, rmu = 0.0f, rdiag_first = 0.0f;
  float rsolref[2] = {0.02f, 1.0f}, rsolimp[5] = {0.9f, 0.95f, 0.001f, 0.5f, 2.0f}; /* copied per row kind: typed loads, no flat pointer */
  float flR = 0.0f, flB = 0.0f; /* friction-loss rows: host-folded R and damping gain */
  /* elliptic rows (CONE): position in the contact and its dim, first row of the contact, friction coefficient of this
   * row (e >= 1), the contact's mu = friction_0 / sqrt(impratio) */
  int ecode = 0, er0 = lane;
  float efri = 0.0f, emu = 0.0f, eR0 = 1.0f, econ_dist = 0.0f, econ_inc = 0.0f;
  int jd = -1, jleg = -1, jdepth = -1;   /* single-entry rows: dof index; contact rows: leg / depth of the body (-1: base) */
  int jleg1 = -1, jdepth1 = -1;          /* robot-robot contacts (SELF variants): chain of the contact's FIRST body, entering J with a minus sign */
  bool internal = false;
  float jsgn = 0.0f;
  bool jcon = false;
  V3 dir = v3(0.0f, 0.0f, 0.0f), w = v3(0.0f, 0.0f, 0.0f);
  if constexpr (!EARLY) fetch_flr();
  if (lane < nfl) {
    rtype = ROW_FRICTION; rfloss = flr_floss; flR = flr_R; flB = flr_B;
    jd = flr_dof; jsgn = 1.0f;
  } else if (lane < nfl + nlim) {
    const int r = lane - nfl, j = W.u2.c.lim_jnt[r], d = 6 + j;
    rtype = ROW_LIMIT; rpos = W.u2.c.lim_dist[r]; rmargin = m.jnt_margin[j]; rdiag = m.dof_invweight0[d];
    rsolref[0] = m.jnt_solref[j][0]; rsolref[1] = m.jnt_solref[j][1];
#pragma unroll
    for (int q = 0; q < 5; q++) rsolimp[q] = m.jnt_solimp[j][q];
    jd = d; jsgn = W.u2.c.lim_side[r];
  } else if (lane < nefc) {
    int c = 0;
    for (int q = 1; q < ncon; q++)
      if (lane >= W.con_row[q]) c = q;
    const int e = lane - W.con_row[c], dim = W.con_dim[c];
    int body = W.con_body[c], body1 = -1;
    if constexpr (GEN) { body1 = GQ_CON_BODY1(body); body = GQ_CON_BODY2(body); }
    const float mu = W.con_mu[c];
    rpos = W.con_dist[c]; rmargin = W.con_inc[c];
    rsolref[0] = W.con_solref[c][0]; rsolref[1] = W.con_solref[c][1];
#pragma unroll
    for (int q = 0; q < 5; q++) rsolimp[q] = W.con_solimp[c][q];
    float tran = m.body_invweight0[body][0], rotw = m.body_invweight0[body][1];
    if constexpr (GEN) if (body1 >= 0) { /* contact between two bodies of the robot: both bodies' weights (mj_makeConstraint) */
      tran += m.body_invweight0[body1][0]; rotw += m.body_invweight0[body1][1];
      internal = true;
      if (body1 > 0) { jleg1 = (body1 - 1) / 3; jdepth1 = (body1 - 1) % 3; }
    }
    /* contact frame (mju_makeFrame): horizontal floor n = z, t1 = y, t2 = -x; box contacts bring their own normal */
    /* floor: n = z, t1 = (cos, sin, 0) as the narrow phase left it (default y; a capsule's axis), t2 = n x t1 */
    const float t1c = W.con_t1[c][0], t1s = W.con_t1[c][1];
    V3 cn = v3(0.0f, 0.0f, 1.0f), ct1 = v3(t1c, t1s, 0.0f), ct2 = v3(-t1s, t1c, 0.0f);
    if constexpr (GEN) if (GQ_BX_WCLS(W)[c] != -1) { cn = ld3(GQ_BX_CONNRM(W) + 3 * c); make_frame(cn, ct1, ct2); } /* world boxes, height field, robot-robot: mju_makeFrame of their own normal */
    dir = cn;
    bool rotational = false;
    if (dim == 1) { rtype = ROW_CONTACT1; rdiag = tran; }
    else if constexpr (CONE) {
      rtype = ROW_ELLIPTIC; ecode = e | (dim << 4); er0 = W.con_row[c];
      econ_dist = rpos; econ_inc = rmargin;
      emu = mu * K.impratio_rs;
      /* torsional / rolling coefficients: same mixing rule as the sliding one (S6), _set_ground_friction overrides
       * floor and feet with [mu, 0.005, 0.0]; clamped at mjMINMU */
      const int cword = W.con_geom[c], code = GEN ? (cword & 0xff) : cword, code1 = GEN ? ((cword >> 8) & 0xff) - 1 : -1;
      const float mu_env = W.mu_env;
      const int rule = code1 >= 0 ? (cword >> 16) & 3 : (code < 4 ? m.foot_fric_rule[code] : m.lg[code - 4].fric_rule);
      float fr[3] = {mu, 0.0f, 0.0f};
#pragma unroll
      for (int q = 1; q < 3; q++) {
        const float ovr = q == 1 ? 0.005f : 0.0f;
        float ff = mu_env >= 0.0f ? ovr : m.floor_friction[q];
        if constexpr (GEN) {
          const int wc = GQ_BX_WCLS(W)[c];
          if (wc >= 0) ff = m.boxcls_friction[wc][q];
          if (code1 >= 0) ff = code1 < 4 ? (mu_env >= 0.0f ? ovr : m.foot_friction[code1][q]) : m.lg[code1 - 4].friction[q]; /* the contact's first geom is a robot geom */
        }
        const float fg = code < 4 ? (mu_env >= 0.0f ? ovr : m.foot_friction[code][q]) : m.lg[code - 4].friction[q];
        fr[q] = fmaxf(1e-5f, rule == 0 ? fmaxf(ff, fg) : (rule == 1 ? ff : fg));
      }
      efri = e == 0 ? 0.0f : (e < 3 ? fr[0] : (e == 3 ? fr[1] : fr[2]));
      rotational = e >= 3;
      const int ax = e % 3; /* 0: n, 1: t1, 2: t2 */
      dir = ax == 0 ? cn : (ax == 1 ? ct1 : ct2);
      rdiag = rotational ? rotw : tran;
      rdiag_first = tran;                                   /* R of the contact's normal row */
      if (e > 0) { rpos = 0.0f; rmargin = 0.0f; }           /* friction rows carry no penetration */
    } else {
      rtype = ROW_PYRAMID;
      const float sgn = (e & 1) ? -mu : mu;
      dir = cn + sgn * ((e >> 1) == 0 ? ct1 : ct2);
      rdiag = tran + mu * mu * tran;
      rdiag_first = rdiag;
      rmu = mu * K.impratio_rs;
    }
    w = cross(ld3(W.con_pos[c]) - v3(0.0f, 0.0f, W.basez), dir); /* O re-read: not kept live across the stages */
    if (rotational) { w = dir; dir = v3(0.0f, 0.0f, 0.0f); } /* torsion / rolling rows act on the angular Jacobian */
    jcon = true;
    if (body > 0) { jleg = (body - 1) / 3; jdepth = (body - 1) % 3; }
  }
  GQ_SUB(W, 2, 1); /* row descriptors */
  float J[SOLVER == 1 ? 1 : GQ_NVD];
  float vel = 0.0f;
  if constexpr (SOLVER == 1) {
    /* Newton: the row goes straight to LDS.  A contact row has at most nine non-zeros - the six base dofs and the dofs of the contact
     * body's own leg down to its depth - so the sweep walks those nine (six wave-uniform reads of cdof, three per-lane gathers) over a
     * zero-filled row instead of all eighteen dofs with a three-way select each (round 4: 324 instructions, 3.6 k cycles for a lone wave);
     * single-entry rows (friction loss, limits) drop their +-1 on top.  Same products in the same order: the values are the old ones. */
    float* Jr = W.u.B[lane];
#pragma unroll
    for (int k = 0; k < GQ_NVD; k++) Jr[k] = 0.0f; /* (rows >= nefc stay all-zero: rtype NONE sets no descriptor) */
    const bool base_on = jcon && !internal; /* robot-robot rows: the base columns cancel */
#pragma unroll
    for (int k = 0; k < 6; k++) {
      const float* sd = W.cdof[k]; /* wave-uniform LDS reads */
      float v = sd[0] * w.x + sd[1] * w.y + sd[2] * w.z + sd[3] * dir.x + sd[4] * dir.y + sd[5] * dir.z;
      v = base_on ? v : 0.0f;
      vel += v * W.qvel[k];
      Jr[k] = v;
    }
    {
      const int k0 = 6 + 3 * (jleg < 0 ? 0 : jleg);
#pragma unroll
      for (int i = 0; i < 3; i++) {
        const float* sd = W.cdof[k0 + i]; /* per-lane gather */
        float v = sd[0] * w.x + sd[1] * w.y + sd[2] * w.z + sd[3] * dir.x + sd[4] * dir.y + sd[5] * dir.z;
        v = (jcon && jleg >= 0 && i <= jdepth) ? v : 0.0f;
        vel += v * W.qvel[k0 + i];
        Jr[k0 + i] = v;
      }
    }
    if (jd >= 0) { Jr[jd] = jsgn; vel += jsgn * W.qvel[jd]; } /* (LDS executes a lane's writes in order: this lands on top of the zero) */
    if constexpr (SELF) if (uniform(W.nself) > 0) { /* wave-uniform, rare: J(second body) - J(first body) at the contact point - the chain of
                                                     * the contact's first body enters with a minus sign (dofs the two chains share cancel) */
      const int k1 = 6 + 3 * (jleg1 < 0 ? 0 : jleg1);
#pragma unroll
      for (int i = 0; i < 3; i++) {
        const float* sd = W.cdof[k1 + i];
        const bool on1 = internal && jleg1 >= 0 && i <= jdepth1;
        const float v1 = on1 ? sd[0] * w.x + sd[1] * w.y + sd[2] * w.z + sd[3] * dir.x + sd[4] * dir.y + sd[5] * dir.z : 0.0f;
        vel -= v1 * W.qvel[k1 + i];
        Jr[k1 + i] -= v1;
      }
    }
  } else {
#pragma unroll
  for (int k = 0; k < GQ_NVD; k++) {
    const float* sd = W.cdof[k]; /* wave-uniform LDS reads */
    float v = sd[0] * w.x + sd[1] * w.y + sd[2] * w.z + sd[3] * dir.x + sd[4] * dir.y + sd[5] * dir.z;
    const bool on_chain = k < 6 ? (jcon && !internal) : (jcon && (k - 6) / 3 == jleg && (k - 6) % 3 <= jdepth); /* robot-robot rows: the base columns cancel */
    v = on_chain ? v : (k == jd ? jsgn : 0.0f);
    vel += v * W.qvel[k];
    if constexpr (SOLVER == 1) W.u.B[lane][k] = v; /* rows >= nefc are all-zero: rtype NONE sets no descriptor */
    else J[k] = v;
  }
  if constexpr (SELF) if (uniform(W.nself) > 0) { /* wave-uniform, rare: J(second body) - J(first body) at the contact point -
                                                     * the chain of the contact's first body enters with a minus sign (dofs the
                                                     * two chains share cancel) */
#pragma unroll
    for (int k = 6; k < GQ_NVD; k++) {
      const float* sd = W.cdof[k];
      const bool on1 = internal && (k - 6) / 3 == jleg1 && (k - 6) % 3 <= jdepth1;
      const float v1 = on1 ? sd[0] * w.x + sd[1] * w.y + sd[2] * w.z + sd[3] * dir.x + sd[4] * dir.y + sd[5] * dir.z : 0.0f;
      vel -= v1 * W.qvel[k];
      if constexpr (SOLVER == 1) W.u.B[lane][k] -= v1;
      else J[k] -= v1;
    }
  }
  }
  GQ_SUB(W, 2, 2); /* J sweep */
  float rR = 1.0f, raref = 0.0f;
  if (rtype == ROW_FRICTION) { rR = flR; raref = -flB * vel; }
  else if (rtype != ROW_NONE) {
    float imp = impedance(rsolimp, rpos, rmargin);
    rR = fmaxf(1e-15f, fdiv((1.0f - imp) * rdiag, imp));
    float dmax = fminf(fmaxf(rsolimp[1], 0.0001f), 0.9999f), K, B;
    if (rsolref[0] > 0.0f) {
      float tc = fmaxf(rsolref[0], 2.0f * h), dr = rsolref[1];
      K = fast_rcp(fmaxf(1e-15f, dmax * dmax * tc * tc * dr * dr));
      B = 2.0f * fast_rcp(fmaxf(1e-15f, dmax * tc));
    } else { K = fdiv(-rsolref[0], fmaxf(1e-15f, dmax * dmax)); B = fdiv(-rsolref[1], fmaxf(1e-15f, dmax)); }
    raref = -B * vel - K * imp * (rpos - rmargin);
    if (rtype == ROW_PYRAMID) { /* Rpy = 2 mu^2 R(first edge); all edges of a condim-3 contact share diagApprox */
      float Rfirst = fmaxf(1e-15f, fdiv((1.0f - imp) * rdiag_first, imp));
      rR = fmaxf(1e-15f, 2.0f * rmu * rmu * Rfirst);
    }
    if constexpr (CONE) if (rtype == ROW_ELLIPTIC) {
      /* R_n from the normal row's impedance (penetration of the contact), friction rows R_j = R_n mu^2 / friction_j^2 */
      const float impn = impedance(rsolimp, econ_dist, econ_inc);
      eR0 = fmaxf(1e-15f, fdiv((1.0f - impn) * rdiag_first, impn));
      if ((ecode & 15) > 0) rR = fmaxf(1e-15f, fdiv(eR0 * emu * emu, efri * efri));
      else rR = eR0;
    }
  }

  GQ_TICK(7); GQ_SUB(W, 2, 3); /* impedance, R, aref */
  const bool active = lane < nefc;
  float b_i = 0.0f;
  int iter = 0;
  /* inspection record of this forward pass (instrumented variant): `solved` = the solver and the accelerations are done */
  auto dump_record = [&](const bool solved, const float raref_, const float rR_, const int rtype_, const int iter_) {
    if constexpr (DBG) if (call.debug && rec_pass && env < Bt.debug_envs) {
    float* D = call.debug + (size_t)env * GQ_DBG_SIZE;
    for (int k = lane; k < 324; k += GQ_WAVE) D[GQ_DBG_M + k] = m_entry(W, k / 18, k % 18);
    if (lane < 18) {
      D[GQ_DBG_BIAS + lane] = W.bias[lane]; D[GQ_DBG_SMOOTH + lane] = W.smooth[lane];
      D[GQ_DBG_QACC_SMOOTH + lane] = W.qacc_smooth[lane]; D[GQ_DBG_QFRC_C + lane] = W.qfrc_c[lane];
      D[GQ_DBG_QACC + lane] = W.qacc[lane];
    }
    if (lane == 0) { D[GQ_DBG_NEFC] = (float)nefc; D[GQ_DBG_NCON] = (float)ncon; D[GQ_DBG_NITER] = (float)iter_; }
    for (int k = 0; k < GQ_NVD; k++) D[GQ_DBG_EFC_J + lane * 18 + k] = W.u.B[lane][k];
    if constexpr (SOLVER == 1) { /* efc_b = J qacc_smooth - aref: only the record wants it, the primal solver never forms it */
      b_i = -raref_;
      for (int k = 0; k < GQ_NVD; k++) b_i += W.u.B[lane][k] * W.qacc_smooth[k];
    }
    D[GQ_DBG_EFC_AREF + lane] = raref_; D[GQ_DBG_EFC_R + lane] = rR_; D[GQ_DBG_EFC_B + lane] = b_i;
    D[GQ_DBG_EFC_FORCE + lane] = W.force[lane]; D[GQ_DBG_EFC_TYPE + lane] = (float)rtype_;
    if (lane < GQ_MAXCON) { D[GQ_DBG_CON_DIST + lane] = lane < ncon ? W.con_dist[lane] : 0.0f; D[GQ_DBG_CON_GEOM + lane] = lane < ncon ? (float)W.con_geom[lane] : -1.0f; }
    if (lane < 12) D[GQ_DBG_FOOT_POS + lane] = W.foot_world[lane / 3][lane % 3];
      }
    (void)solved;
  };
  if constexpr (DBG) if (call.forward == 1) { /* mj_step1: position and velocity stages are done (constraint rows incl. aref) */
    if constexpr (SOLVER == 1) wave_barrier();
    dump_record(false, raref, rR, rtype, 0);
    return 0;
  }

  float pgs_keep = 0.0f;
  int hint_out = 0; /* solver load of this step: the env's issue-priority hint for its next one */
  if constexpr (SOLVER == 1) {
    /* ================================================================ S8/S9 (Newton): primal solve, no dual operator */
    wave_barrier(); /* the J rows are in LDS (S7, over the dead u.dyn) for the Hessian assembly */
    GQ_TICK(8);
    const EllRow ell = {ecode, er0, efri, emu, fast_rcp(eR0)};
    if constexpr (!EARLY) fetch_hent();
    /* a contact between two different legs couples them in the Hessian M + J'DJ, which then no longer has M's tree
     * sparsity: such an env takes the dense Newton step */
    const bool xrow = SELF && internal && jleg1 >= 0 && jleg >= 0 && jleg1 != jleg && K.self_cut != 3;
    const bool xleg = SELF && ballot(xrow) != 0;
    if constexpr (DBG && SELF) if (timing && lane == 0) call.debug[(size_t)env * GQ_DBG_SIZE + GQ_DBG_TIMER + 31] = (float)(uniform(W.nself) + (xleg ? 100 : 0));
    int fl_row_nw = Drec.fl_row;
    if constexpr (!EARLY) { int l_ = ld; opaque(l_); fl_row_nw = m.dof_rec[l_].fl_row; } /* (a fresh load instead of a register kept - spilled - since the kernel's first lines) */
    const float fN = newton_solve<DBG, CONE>(W, K, fl_row_nw, hent_pre, rtype, rR, raref, rfloss, nefc, nfl, nfl + nlim, iter,
                                  timing ? call.debug + (size_t)env * GQ_DBG_SIZE + GQ_DBG_TIMER : nullptr, ell, prio_hint, xrow,
                                  xleg ? uniform(W.con_row[ncon - uniform(W.nself)]) : -1);
    /* a coupled-leg Newton step (Sherman-Morrison / dense) is the most expensive thing a wave can do, and leg-leg contacts
     * persist over several steps: such an env keeps top issue priority */
    /* (stored with the epilogue's batch of pointers, below) */
    if constexpr (CONE) hint_out = iter >= 8 ? 3 : (iter >= 6 ? 2 : (iter >= 4 ? 1 : 0)); /* see newton_solve */
    else hint_out = (xleg && iter >= 2) ? 3 : (iter < 2 ? 0 : (iter > 2 ? 3 : 2));
    W.force[lane] = active ? fN : 0.0f;
    wave_barrier();
  } else {
  /* PGS with world geoms / self-collision: the contacts' normals and world classes (48 words, read again by S11) sit where the factors
   * go - one register per lane carries them across the solve */
  float keep_scr = 0.0f;
  if constexpr (BOXES || SELF) {
    wave_barrier(); /* S7 has read them */
    if (lane < 36) keep_scr = GQ_BX_CONNRM(W)[lane];
    else if (lane < 48) keep_scr = __builtin_bit_cast(float, GQ_BX_WCLS(W)[lane - 36]);
    wave_barrier();
    factor_tree_both(W, m.dof_damping, h);
  }
  /* ================================================================ S8: B = M^-1 J' (lane-parallel), A = J B' + R */
  float A[GQ_MAXEFC];
  float diag = 0.0f; /* A_ii = J_i . B_i */
  {
    float x[GQ_NVD];
#pragma unroll
    for (int k = 0; k < GQ_NVD; k++) x[k] = (lane == 63) ? W.smooth[k] : J[k];
    solve_tree(W, 0, x);
#pragma unroll
    for (int k = 0; k < GQ_NVD; k++) diag += J[k] * x[k];
    if (lane == 63) {
#pragma unroll
      for (int k = 0; k < GQ_NVD; k++) W.qacc_smooth[k] = x[k];
    }
    /* rows of B are exchanged through LDS (they overlay the spatial-dynamics scratch, dead by now) */
    wave_barrier();
#pragma unroll
    for (int k = 0; k < GQ_NVD; k++) W.u.B[lane][k] = x[k];
    wave_barrier();
  }
#pragma unroll
  for (int j = 0; j < GQ_MAXEFC; j++) {
    float sacc = 0.0f;
    if (j < nefc) { /* wave-uniform */
#pragma unroll
      for (int k = 0; k < GQ_NVD; k++) sacc += J[k] * W.u.B[j][k];
    }
    A[j] = sacc;
    sched_fence(); /* one row of B in flight at a time: keeps the register pressure of the unrolled loop flat */
  }
  b_i = -raref;
  float jar_w = -raref;
#pragma unroll
  for (int k = 0; k < GQ_NVD; k++) { b_i += J[k] * W.qacc_smooth[k]; jar_w += J[k] * W.warm[k]; }
  /* J rows take B's place in LDS (needed again for J'f); their registers are free during the PGS sweeps */
  wave_barrier();
#pragma unroll
  for (int k = 0; k < GQ_NVD; k++) W.u.B[lane][k] = (lane < nefc) ? J[k] : 0.0f;

  GQ_TICK(8);
  /* ================================================================ S9: PGS on  min 1/2 f'(A+R)f + f'b */
  float lo = 0.0f, hi = 3.0e38f;
  if (rtype == ROW_FRICTION) { lo = -rfloss; hi = rfloss; }
  if (!active) { lo = 0.0f; hi = 0.0f; }
  const float ARii = active ? diag + rR : 1.0f;
  const float invd = 1.0f / ARii;
  /* warm start: primal force law at qacc_warmstart (mj_constraintUpdate) */
  float f = 0.0f;
  if (active) {
    if (rtype == ROW_FRICTION) f = med3(-jar_w / rR, -rfloss, rfloss);
    else f = jar_w < 0.0f ? -jar_w / rR : 0.0f;
  }
  /* residual r = (A+R) f + b ; dual cost decides whether the warm start is kept */
  float r = b_i + rR * f;
#pragma unroll
  for (int j = 0; j < GQ_MAXEFC; j++) {
    if (j < nefc) r += A[j] * bcast(f, j);
  }
  {
    float cost = wave_sum(active ? 0.5f * f * (r - b_i) + f * b_i : 0.0f);
    if (cost > 0.0f) { f = 0.0f; r = b_i; }
  }
  if (!active) r = 0.0f;
  const float scale = K.nw_scale;
  /* fold R into the diagonal: lane i's own column entry becomes (A+R)_ii, so the residual update below is one
   * uniform FMA for every lane */
#pragma unroll
  for (int j = 0; j < GQ_MAXEFC; j++) A[j] = (j == lane) ? ARii : A[j];
  for (; iter < K.iterations; iter++) {
    float imp_acc = 0.0f;
    int lane_s = lane;
    opaque(lane_s); /* keeps the 63 (lane == i) predicates from being hoisted out of the sweep loop as live masks */
#pragma unroll
    for (int i0 = 0; i0 < GQ_MAXEFC + 1; i0 += 4) {
      if (i0 < nefc) { /* wave-uniform; rows past nefc inside a group are inert (lo = hi = 0, f = 0) */
#pragma unroll
        for (int i = i0; i < i0 + 4 && i < GQ_MAXEFC; i++) {
          const float fn = med3(f - r * invd, lo, hi);
          const float delta = fn - f;
          if (lane_s == i) {
            imp_acc -= delta * (0.5f * delta * ARii + r);
            f = fn;
          }
          r += A[i] * bcast(delta, i);
        }
      }
    }
    float improvement = wave_sum(imp_acc);
    if (improvement * scale < K.tolerance) { iter++; break; }
  }
  W.force[lane] = active ? f : 0.0f;
  wave_barrier();
  pgs_keep = keep_scr;
  }
  GQ_TICK(9); GQ_SUB(W, 2, 4); /* solver */
  /* ================================================================ S10: accelerations and integration */
  if constexpr (SOLVER == 1) {
    /* qacc and qfrc_constraint (= M (qacc - qacc_smooth)) come out of the Newton solve; only the Euler system
     * (M + h D) qacc_int = qfrc_smooth + qfrc_constraint is left */
    W.act[ld] = W.smooth[ld] + W.qfrc_c[ld];
    wave_barrier();
    solve_tree_stored(GQ_EULER_FLEG(W), GQ_EULER_FBASE(W), W.act, W.qacc_int); /* factor of M + h diag(damping): left in LDS by the solver's first elimination */
  } else {
  if (lane < GQ_NVD) { /* qfrc_constraint = J' f: four independent partial sums keep the LDS reads pipelined */
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    int i = 0;
    for (; i + 4 <= nefc; i += 4) {
      s0 += W.u.B[i][lane] * W.force[i]; s1 += W.u.B[i + 1][lane] * W.force[i + 1];
      s2 += W.u.B[i + 2][lane] * W.force[i + 2]; s3 += W.u.B[i + 3][lane] * W.force[i + 3];
    }
    for (; i < nefc; i++) s0 += W.u.B[i][lane] * W.force[i];
    W.qfrc_c[lane] = (s0 + s1) + (s2 + s3);
  }
  wave_barrier();
  { /* qacc = qacc_smooth + M^-1 qfrc_c ;  Euler: (M + h D) qacc_int = qfrc_smooth + qfrc_c.  Even lanes solve the
     * first system, odd lanes the second: one pass over each factor, every lane busy */
    float x[GQ_NVD], y[GQ_NVD];
#pragma unroll
    for (int k = 0; k < GQ_NVD; k++) { x[k] = W.qfrc_c[k]; y[k] = x[k] + W.smooth[k]; }
    solve_tree(W, 0, x);
    solve_tree(W, 1, y);
    if (lane < GQ_NVD) {
      float qa = 0.0f, qi = 0.0f;
#pragma unroll
      for (int k = 0; k < GQ_NVD; k++) { qa = (k == lane) ? x[k] : qa; qi = (k == lane) ? y[k] : qi; }
      W.qacc[lane] = W.qacc_smooth[lane] + qa;
      W.qacc_int[lane] = qi;
    }
  }
  wave_barrier();
  if constexpr (BOXES || SELF) { /* the factors are dead: the contact normals / world classes return for S11 */
    if (lane < 36) GQ_BX_CONNRM(W)[lane] = pgs_keep;
    else if (lane < 48) GQ_BX_WCLS(W)[lane - 36] = __builtin_bit_cast(int32_t, pgs_keep);
    wave_barrier();
  }

  }
  dump_record(true, raref, rR, rtype, iter);
  if constexpr (DBG) if (fwd_only) { /* mj_forward: the acceleration is the result; nothing is integrated, no observation row */
    if (lane < GQ_NVD) gptr(a.qacc)[(size_t)env * 18 + lane] = W.qacc[lane];
    return 0;
  }

  GQ_TICK(10); GQ_SUB(W, 2, 5); /* Euler system */
  /* ---- the epilogue's pointers and batch / model scalars: ONE batch of scalar loads, pinned (round 5; every store below used to fetch
   * its pointer - and the time step, five times - in front of itself: ~25 exposed scalar-cache round trips between here and the end) */
  double* e_qpos = a.qpos; float* e_qvel = a.qvel; float* e_qacc = a.qacc; float* e_warm = a.warm; float* e_obs = a.obs; float* e_reward = a.reward;
  uint8_t* e_term = a.terminated; uint8_t* e_trunc = a.truncated; uint8_t* e_inval = a.invalid_contact; uint8_t* e_pending = a.pending;
  int32_t* e_dropped = a.contacts_dropped; float* e_imu_bias = a.imu_bias; float* e_contacts = a.contacts; int32_t* e_h9 = a.h9;
  float* e_obs_seq = call.obs_seq; uint8_t* e_hint = a.load_hint;
  int od = Bt.obs_dim, obs_need = Bt.obs_need, imu_en = Bt.imu_enabled;
  double tlim0 = m.terrain_limits[0], tlim1 = m.terrain_limits[1], tlim2 = m.terrain_limits[2], tlim3 = m.terrain_limits[3];
  pin(e_qpos); pin(e_qvel); pin(e_qacc); pin(e_warm); pin(e_obs); pin(e_reward); pin(e_term); pin(e_trunc); pin(e_inval); pin(e_pending);
  pin(e_dropped); pin(e_imu_bias); pin(e_contacts); pin(e_h9); pin(e_obs_seq); pin(e_hint); pin(od); pin(obs_need); pin(imu_en);
  pin(tlim0); pin(tlim1); pin(tlim2); pin(tlim3);
  /* the output layout (column -> canonical scalar) of this lane's columns: fetched now, used by the gather at the end */
  int omap[4], s3k[3];
#pragma unroll
  for (int i = 0; i < 4; i++) { const int k = lane + GQ_WAVE * i; omap[i] = k < od ? Bt.obs_map[k] : 0; }
  int lane_s3 = lane;
  if constexpr (!EARLY) opaque(lane_s3); /* (a fresh load through an opaque index: otherwise S3's copy of the same words is kept - in scratch memory - across the solver) */
#pragma unroll
  for (int p = 0; p < 3; p++) s3k[p] = (SOLVER == 1 && (obs_need & GQ_NEED_ENERGY)) ? m.s3_ent[p][lane_s3] : 0; /* the energy sums walk M's stored entries */
  /* IMU ground truth (mj_sensorAcc / mj_sensorVel of this forward pass: OLD pose and velocity, this step's qacc).
   * accelerometer = site-frame acceleration of the site point minus gravity; gyro = site-frame angular velocity */
  const bool imu_on = e_imu_bias != nullptr && imu_en;
  if (imu_on && lane == 0) {
    const GQ_MODEL GqDevBatch& B = Bt;
    Q4 qo = {W.qb[0], W.qb[1], W.qb[2], W.qb[3]};
    float Ro[9];
    q2mat(Ro, qnormalize(qo));
    const V3 wb = v3(W.qvel[3], W.qvel[4], W.qvel[5]);
    const V3 ww = matvec(Ro, wb), aw = matvec(Ro, v3(W.qacc[3], W.qacc[4], W.qacc[5]));
    const V3 r = matvec(Ro, ld3(B.imu_pos));
    V3 ap = v3(W.qacc[0], W.qacc[1], W.qacc[2]) + cross(aw, r) + cross(ww, cross(ww, r));
    ap.z -= K.gravity_z;
    const V3 acc_s = matTvec(B.imu_mat, matTvec(Ro, ap)), gyr_s = matTvec(B.imu_mat, wb);
    W.warm[0] = acc_s.x; W.warm[1] = acc_s.y; W.warm[2] = acc_s.z; W.warm[3] = gyr_s.x; W.warm[4] = gyr_s.y; W.warm[5] = gyr_s.z;
  }

  /* mj_checkAcc: a non-finite or absurd acceleration (|qacc| >= 1e10, MuJoCo's mjMAXVAL) means the simulation diverged.
   * MuJoCo resets the data and warns; here the env is frozen for this step (zero acceleration) and flagged terminated +
   * truncated, so that an auto-resetting batch re-spawns it and a NaN never reaches the next step */
  const float qa_raw = W.qacc[ld], qi_raw = W.qacc_int[ld]; /* lane = dof, mirror lanes */
  const bool diverged = ballot(!(fabsf(qa_raw) < 1e10f && fabsf(qi_raw) < 1e10f)) != 0;
  /* semi-implicit Euler (mj_Euler): velocity with the damped system, then positions with the new velocity */
  float vnew = 0.0f;
  {
    const float qa = diverged ? 0.0f : qa_raw, qi = diverged ? 0.0f : qi_raw;
    if (diverged) { W.qacc[ld] = 0.0f; W.qacc_int[ld] = 0.0f; } /* wave-uniform, rare */
    vnew = W.qvel[ld] + h * qi;
    vnew = fabsf(vnew) < 1e10f ? vnew : 0.0f;
    gptr(e_qvel)[(size_t)env * 18 + ld] = vnew;
    gptr(e_qacc)[(size_t)env * 18 + ld] = qa;
    gptr(e_warm)[(size_t)env * 18 + ld] = qa;
  }
  /* base x,y stay in f64 (uniform: every lane reads the same two words) and never enter fp32 arithmetic; they wait
   * in LDS since S0, so they do not occupy registers across the solver */
  const double bx_d = W.bxy[0], by_d = W.bxy[1];
  wave_barrier();
  W.qvel[ld] = vnew; /* new qvel; old one is not needed any more */
  wave_barrier();
  /* positions */
  const double bxn_d = bx_d + (double)h * (double)W.qvel[0], byn_d = by_d + (double)h * (double)W.qvel[1];
  const float znew = W.basez + h * W.qvel[2];
  Q4 qn;
  {
    V3 w = v3(W.qvel[3], W.qvel[4], W.qvel[5]);
    float n = fast_sqrt(dot(w, w));
    Q4 qraw = {W.qb[0], W.qb[1], W.qb[2], W.qb[3]};
    const Q4 qbase = qnormalize(qraw);
    qn = qbase;
    /* mju_quatIntegrate starts from the raw (un-normalised) qpos quaternion; it was normalised above, the
     * difference is removed by the normalisation that follows */
    if (n > 1e-15f) {
      float ang = h * n, s, c;
      sincos_small(0.5f * ang, s, c);
      s = fdiv(s, n);
      Q4 qr = {c, w.x * s, w.y * s, w.z * s};
      qn = qmul(qbase, qr);
    }
    qn = qnormalize(qn);
  }
  /* the qpos row: lane = column (mirror lanes), ONE store - base x, y (f64), z, the quaternion, the joint angles picked by selects */
  const int l19 = lane < 19 ? lane : 18, jq = l19 < 7 ? 0 : l19 - 7;
  const float qjn = W.qj[jq] + h * W.qvel[6 + jq];
  const float qcol = l19 < 3 ? znew : (l19 == 3 ? qn.w : (l19 == 4 ? qn.x : (l19 == 5 ? qn.y : (l19 == 6 ? qn.z : qjn)))); /* columns 2 .. 18 */
  {
    const double qd = l19 == 0 ? bxn_d : (l19 == 1 ? byn_d : (double)qcol);
    gptr(e_qpos)[(size_t)env * 19 + l19] = qd;
  }

  GQ_TICK(11); GQ_SUB(W, 2, 6); /* integration + state stores */
  /* ================================================================ S11: observations (new qpos/qvel, old kinematics) */
  float Rn[9];
  q2mat(Rn, qn);
  V3 vlin = v3(W.qvel[0], W.qvel[1], W.qvel[2]), wloc = v3(W.qvel[3], W.qvel[4], W.qvel[5]);
  float* ob = W.u.obs;
  /* observables nobody asked for are not computed (obs_need: groups of canonical scalars the output layout refers to) */
  ob[OB_QPOS + l19] = l19 == 0 ? (float)bxn_d : (l19 == 1 ? (float)byn_d : qcol); /* the same row in fp32: one store */
  if (obs_need & GQ_NEED_BASE) {
  /* scipy as_euler('xyz') of the new orientation */
  float sy = fminf(fmaxf(-Rn[6], -1.0f), 1.0f);
  float e0, e1 = asinf(sy), e2, cyaw, syaw;
  if (fabsf(sy) < 0.9999999f) { e0 = atan2_fast(Rn[7], Rn[8]); e2 = atan2_fast(Rn[3], Rn[0]); syaw = Rn[3]; cyaw = Rn[0]; }
  else { e0 = 0.0f; e2 = atan2_fast(-Rn[1], Rn[4]); syaw = -Rn[1]; cyaw = Rn[4]; }
  { /* cos / sin of the yaw angle straight from the atan2 arguments */
    const float hyp2 = syaw * syaw + cyaw * cyaw, inv = hyp2 > 0.0f ? fast_rsqrt(hyp2) : 0.0f;
    syaw *= inv; cyaw = hyp2 > 0.0f ? cyaw * inv : 1.0f;
  }
  V3 cmdl = v3(W.cmd[0], W.cmd[1], W.cmd[2]);
  V3 tl = v3(cyaw * cmdl.x - syaw * cmdl.y, syaw * cmdl.x + cyaw * cmdl.y, cmdl.z);
  V3 ta = v3(0.0f, 0.0f, W.cmd[3]);
  V3 acc3 = v3(W.qacc[0], W.qacc[1], W.qacc[2]);
  { /* (every lane stores the same words: the values are wave-uniform) */
    ob[OB_BASE_POS] = (float)bxn_d; ob[OB_BASE_POS + 1] = (float)byn_d; ob[OB_BASE_POS + 2] = znew;
    st3(ob + OB_LIN_VEL, vlin); st3(ob + OB_LIN_VEL_ERR, tl - vlin); st3(ob + OB_LIN_ACC, acc3);
    V3 ww = matvec(Rn, wloc);
    st3(ob + OB_ANG_VEL, ww); st3(ob + OB_ANG_VEL_ERR, ta - ww);
    ob[OB_EULER] = e0; ob[OB_EULER + 1] = e1; ob[OB_EULER + 2] = e2;
    ob[OB_QUAT] = qn.w; ob[OB_QUAT + 1] = qn.x; ob[OB_QUAT + 2] = qn.y; ob[OB_QUAT + 3] = qn.z;
#pragma unroll
    for (int k = 0; k < 9; k++) ob[OB_SO3 + k] = Rn[k];
    st3(ob + OB_GRAV_B, matTvec(Rn, v3(0.0f, 0.0f, -1.0f)));
    V3 vb = matTvec(Rn, vlin);
    st3(ob + OB_LIN_VEL_B, vb); st3(ob + OB_LIN_VEL_ERR_B, matTvec(Rn, tl) - vb); st3(ob + OB_LIN_ACC_B, matTvec(Rn, acc3));
    st3(ob + OB_ANG_VEL_B, wloc); st3(ob + OB_ANG_VEL_ERR_B, matTvec(Rn, ta) - wloc);
  }
  }
  GQ_SUB(W, 2, 7); /* base observables */
  ob[OB_QVEL + ld] = W.qvel[ld];
  ob[OB_TAU + lj] = W.ctrl[lj]; ob[OB_QVEL_JS + lj] = W.qvel[6 + lj];
  ob[OB_QPOS_JS + jq] = qjn;
  /* kinetic energy 1/2 v'Mv and work (M qacc).v with the OLD mass matrix, NEW velocity, qacc of this step */
  if (obs_need & GQ_NEED_ENERGY) {
    float ke_part = 0.0f, wk_part = 0.0f;
    if constexpr (SOLVER == 1) {
      /* v'Mv summed over the STORED entries of M, lane = entry (S3's table: three per lane; a leg's off-diagonal entries are stored once
       * and count twice, the base block is stored in full), and (M qacc).v with  M qacc = qfrc_smooth + qfrc_constraint,  which S10 left in
       * W.act for the Euler system - no matrix-vector product at all (round 4: two 27-term row products per dof lane behind two
       * divergent branches, 2.9 k cycles for a lone wave).  Equal to the explicit products up to the solver's residual. */
      const float* M0 = &W.Mc[0][0];
#pragma unroll
      for (int p = 0; p < 3; p++) {
        const int ent = s3k[p], dd = ent & 0xff, sa = (ent >> 8) & 0xff, e = lane_s3 + GQ_WAVE * p; /* (opaque in the elliptic variants: S3's copy of this address was kept in scratch) */
        const bool counts = (p < 2 || lane < 144 - 2 * GQ_WAVE) && (ent >> 24) != 0; /* (the table's slots past entry 143 repeat it) */
        const float coef = counts ? ((p < 2 && (p == 0 || lane < 108 - GQ_WAVE) && dd != sa) ? 1.0f : 0.5f) : 0.0f;
        ke_part += coef * M0[e < 143 ? e : 143] * W.qvel[dd] * W.qvel[sa];
      }
      wk_part = (lane < GQ_NVD && !diverged) ? W.act[ld] * W.qvel[ld] : 0.0f; /* (a frozen env: qacc was zeroed) */
    } else if (lane < GQ_NVD) {
      const float mv = mul_m_row(W, W.qvel, lane), ma = mul_m_row(W, W.qacc, lane);
      ke_part = 0.5f * W.qvel[lane] * mv; wk_part = ma * W.qvel[lane];
    }
    float ke = wave_sum(ke_part), wk = wave_sum(wk_part);
    ob[OB_KE] = ke; ob[OB_WORK] = wk;
  }
  GQ_SUB(W, 2, 8); /* joint observables + energy */
  /* feet: lane k < 4 = foot k in FL FR RL RR order */
  if (obs_need & (GQ_NEED_FEET | GQ_NEED_CONTACT)) { /* wave-uniform; lane = foot, mirror lanes */
    const int leg = FRec.leg, body = 3 + 3 * leg;
    V3 pw = ld3(W.foot_world[lq]); /* relative to the OLD base x/y */
    /* spatial velocity of the calf with old cdof and new qvel (J_old * qvel_new) */
    float sv[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int d = 0; d < 6; d++)
#pragma unroll
      for (int k = 0; k < 6; k++) sv[k] += W.cdof[d][k] * W.qvel[d];
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const int d = 6 + 3 * leg + i;
#pragma unroll
      for (int k = 0; k < 6; k++) sv[k] += W.cdof[d][k] * W.qvel[d];
    }
    V3 fv = ld3(sv + 3) + cross(ld3(sv), pw - v3(0.0f, 0.0f, W.basez));
    /* world position of the foot: old base x/y + relative.  feet_pos:base uses the NEW base pose (quirk B3) */
    V3 pworld = v3((float)(bx_d + (double)pw.x), (float)(by_d + (double)pw.y), pw.z);
    V3 prel_new = v3(pw.x - h * W.qvel[0], pw.y - h * W.qvel[1], pw.z - znew);
    V3 fvr = fv - vlin - cross(wloc, prel_new); /* quirk B4: body-frame omega used as world-frame */
    /* feet_contact_state (quadruped_env.py:836-855): every world contact of the foot's BODY (the calf: foot sphere
     * and calf link geom alike) sets the state and adds its force */
    V3 cf = v3(0.0f, 0.0f, 0.0f);
    float cs = 0.0f;
    for (int c = 0; c < ((obs_need & GQ_NEED_CONTACT) ? ncon : 0); c++) { /* wave-uniform trip count; the contact's words are wave-uniform
                                                                            * reads, the foot's share is a select (no divergent branch) */
      const bool mine = W.con_body[c] == body;
      const int r0 = W.con_row[c], dimc = W.con_dim[c];
      float fn, ft1, ft2; /* contact-frame force: mj_contactForce */
      if constexpr (CONE) {
        const float f0 = W.force[r0], f1 = W.force[r0 + 1 < 63 ? r0 + 1 : 63], f2 = W.force[r0 + 2 < 63 ? r0 + 2 : 63];
        fn = f0; ft1 = dimc == 1 ? 0.0f : f1; ft2 = dimc == 1 ? 0.0f : f2;
      } else { /* mju_decodePyramid */
        const float f0 = W.force[r0], f1 = W.force[r0 + 1 < 63 ? r0 + 1 : 63], f2 = W.force[r0 + 2 < 63 ? r0 + 2 : 63], f3 = W.force[r0 + 3 < 63 ? r0 + 3 : 63], mu = W.con_mu[c];
        fn = dimc == 1 ? f0 : f0 + f1 + f2 + f3; ft1 = dimc == 1 ? 0.0f : mu * (f0 - f1); ft2 = dimc == 1 ? 0.0f : mu * (f2 - f3);
      }
      int wcls = -1;
      if constexpr (GEN) wcls = uniform(GQ_BX_WCLS(W)[c]);
      V3 add;
      if (wcls != -1) { /* wave-uniform: frame' * f with the contact's own frame */
        const V3 cn = ld3(GQ_BX_CONNRM(W) + 3 * c);
        V3 ct1, ct2;
        make_frame(cn, ct1, ct2);
        add = fn * cn + ft1 * ct1 + ft2 * ct2;
      } else { /* floor: n = z, t1 = (cos, sin, 0), t2 = (-sin, cos, 0) */
        const float t1c = W.con_t1[c][0], t1s = W.con_t1[c][1];
        add = v3(ft1 * t1c - ft2 * t1s, ft1 * t1s + ft2 * t1c, fn);
      }
      cs = mine ? 1.0f : cs;
      cf = v3(mine ? cf.x + add.x : cf.x, mine ? cf.y + add.y : cf.y, mine ? cf.z + add.z : cf.z);
    }
    if ((W.foot_touch >> lq) & 1) cs = 1.0f; /* contact detected but dropped by the row budget */
    /* slot of this foot in legs_order-dependent observables is resolved by obs_map; canonical order = FL FR RL RR */
    st3(ob + OB_FEET_POS + 3 * lq, pworld);
    st3(ob + OB_FEET_POS_B + 3 * lq, matTvec(Rn, prel_new));
    st3(ob + OB_FEET_VEL + 3 * lq, fv);
    st3(ob + OB_FEET_VEL_REL + 3 * lq, fvr);
    st3(ob + OB_FEET_VEL_B + 3 * lq, matTvec(Rn, fv));
    st3(ob + OB_FEET_VEL_REL_B + 3 * lq, matTvec(Rn, fvr));
    ob[OB_CONTACT_STATE + lq] = cs;
    st3(ob + OB_CONTACT_F + 3 * lq, cf);
    st3(ob + OB_CONTACT_F_B + 3 * lq, matTvec(Rn, cf));
  }
  wave_barrier();
  GQ_SUB(W, 2, 9); /* feet + contact forces */
  /* IMU.step (sensors/imu.py:102-139): noise ~ N(0, sigma), bias += N(0, rate), measurement = truth + bias + noise.
   * lanes 0-11 draw: acc noise xyz, acc bias step xyz, gyro noise xyz, gyro bias step xyz */
  if (imu_on) {
    const GQ_MODEL GqDevBatch& B = Bt;
    float z = 0.0f;
    if (lane < 12) {
      const uint32_t stepc = (uint32_t)W.step_old, epi = a.episode_ro ? (uint32_t)ldv<PUB>(a.episode_ro + env) : 0u;
      z = philox_normal((uint32_t)lane, stepc, (uint32_t)env, 0x1a70u ^ (epi << 8), B.imu_seed_lo, B.imu_seed_hi);
      const int grp = lane / 3;
      z *= grp == 0 ? B.imu_acc_noise : (grp == 1 ? B.imu_acc_bias_rate : (grp == 2 ? B.imu_gyro_noise : B.imu_gyro_bias_rate));
    }
    if (lane < 12) W.warm[6 + lane] = z;
    wave_barrier();
    if (lane < 6) { /* lanes 0-2: accelerometer axes, lanes 3-5: gyro axes */
      const int g = lane / 3, ax = lane % 3;
      const float noise = W.warm[6 + 6 * g + ax], dbias = W.warm[6 + 6 * g + 3 + ax];
      const float bias = ldv<PUB>(e_imu_bias + (size_t)env * 6 + lane) + dbias;
      gptr(e_imu_bias)[(size_t)env * 6 + lane] = bias;
      const int o = g == 0 ? OB_IMU_ACC : OB_IMU_GYRO;
      ob[o + ax] = W.warm[lane] + bias + noise; ob[o + 3 + ax] = noise; ob[o + 6 + ax] = bias;
    }
  } else ob[OB_IMU_ACC + ld] = 0.0f;
  /* termination (quadruped_env.py:283-285): non-foot contact with the ground, or base outside the terrain */
  int terminated = 0;
  {
    const bool oob = bxn_d > tlim0 || bxn_d < tlim1 || byn_d > tlim2 || byn_d < tlim3;
    terminated = W.invalid || oob || diverged;
    { /* (every lane stores the same words) */
      if (pass == 0) { /* the flags of the user's step survive an in-kernel auto-reset */
        gptr(e_inval)[env] = (uint8_t)W.invalid;
        gptr(e_term)[env] = (uint8_t)terminated;
        gptr(e_trunc)[env] = (uint8_t)diverged;
      } else if (pass == 2) { /* reset() reports no termination (quadruped_env.py:406 returns the observation only) */
        gptr(e_inval)[env] = 0; gptr(e_term)[env] = 0; gptr(e_trunc)[env] = 0;
      }
      if (e_pending) gptr(e_pending)[env] = (uint8_t)(pass == 0 ? terminated : 0);
      if (e_dropped) gptr(e_dropped)[env] = W.ndrop;
      if (SOLVER == 1 && e_hint) gptr(e_hint)[env] = (uint8_t)hint_out;
      gptr(e_reward)[env] = 0.0f;
      if (pass != 0 && a.friction && a.friction_next) gptr(const_cast<float*>(a.friction))[env] = ldv<PUB>(a.friction_next + env); /* (written by this wave's own reset_wave) */
    }
  }
  /* the contact row (gq_batch_set_outputs): mjData.contact[] with mj_contactForce, lane = contact */
  static_assert(GQ_CON_MAX == GQ_MAXCON, "a contact row holds GQ_CON_MAX records: one per solver contact");
  static_assert(GQ_DYN_MB == GQ_DYN_MC + sizeof(WaveMem::Mc) / 4 && GQ_DYN_BIAS == GQ_DYN_MB + sizeof(WaveMem::Mb) / 4 && GQ_DYN_XPOS == GQ_DYN_BIAS + GQ_NVD &&
                GQ_DYN_XMAT == GQ_DYN_XPOS + sizeof(WaveMem::xpos) / 4 && GQ_DYN_FOOT == GQ_DYN_XMAT + sizeof(WaveMem::xmat) / 4 &&
                GQ_DYN_STRIDE >= GQ_DYN_FOOT + sizeof(WaveMem::foot_world) / 4, "dynamics-row offsets follow the WaveMem field sizes");
  if (e_contacts && rec_pass) { /* wave-uniform */
    GQ_GLOBAL float* Cn = gptr(e_contacts) + (size_t)env * GQ_CON_STRIDE;
    if (lane == 0) { Cn[0] = (float)ncon; Cn[1] = (float)nefc; Cn[2] = (float)iter; }
    if (lane >= ncon && lane < GQ_CON_MAX) { /* no contact: an all-zero record (gq_contact_force returns zeros for it) */
      GQ_GLOBAL float* R = Cn + 8 + lane * GQ_CON_REC;
#pragma unroll
      for (int q = 0; q < GQ_CON_REC; q++) R[q] = 0.0f;
    }
    if (lane < ncon) {
      const int c = lane, r0 = W.con_row[c], dimc = W.con_dim[c];
      GQ_GLOBAL float* R = Cn + 8 + c * GQ_CON_REC;
      const int cword = W.con_geom[c], it2 = GEN ? (cword & 0xff) : cword, it1 = GEN ? ((cword >> 8) & 0xff) - 1 : -1;
      R[0] = it1 >= 0 ? (float)m.item_geomid[it1] : -1.0f;
      R[1] = (float)m.item_geomid[it2];
      R[2] = W.con_dist[c];
      R[3] = W.con_pos[c][0]; R[4] = W.con_pos[c][1]; R[5] = W.con_pos[c][2];
      V3 cn = v3(0.0f, 0.0f, 1.0f), ct1 = v3(W.con_t1[c][0], W.con_t1[c][1], 0.0f), ct2 = v3(-W.con_t1[c][1], W.con_t1[c][0], 0.0f);
      if constexpr (GEN) if (GQ_BX_WCLS(W)[c] != -1) { cn = ld3(GQ_BX_CONNRM(W) + 3 * c); make_frame(cn, ct1, ct2); }
      R[6] = cn.x; R[7] = cn.y; R[8] = cn.z; R[9] = ct1.x; R[10] = ct1.y; R[11] = ct1.z; R[12] = ct2.x; R[13] = ct2.y; R[14] = ct2.z;
      R[15] = (float)dimc;
      float f6[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
      if (dimc == 1) f6[0] = W.force[r0];
      else if constexpr (CONE) {
#pragma unroll
        for (int q = 0; q < 6; q++) f6[q] = q < dimc ? W.force[r0 + q < 63 ? r0 + q : 63] : 0.0f;
      } else { /* mju_decodePyramid */
        const float f0 = W.force[r0], f1 = W.force[r0 + 1], f2 = W.force[r0 + 2], f3 = W.force[r0 + 3], mu = W.con_mu[c];
        f6[0] = f0 + f1 + f2 + f3; f6[1] = mu * (f0 - f1); f6[2] = mu * (f2 - f3);
      }
#pragma unroll
      for (int q = 0; q < 6; q++) R[16 + q] = f6[q];
      R[22] = W.con_mu[c]; R[23] = 0.0f;
    }
  }
  GQ_TICK(12); GQ_SUB(W, 2, 10); /* IMU, termination, flags */
  /* gather to the requested observation layout: coalesced row write */
  {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int k = lane + GQ_WAVE * i;
      if (k < od) {
        const float val = ob[omap[i]];
        if constexpr (PUB) st_pub(e_obs + (size_t)env * od + k, val);
        else gptr(e_obs)[(size_t)env * od + k] = val;
        if (e_obs_seq) gptr(e_obs_seq)[(size_t)env * od + k] = val; /* persistent rollout: the step's own row of the sequence */
      }
    }
  }
  GQ_SUB(W, 2, 11); /* gather */
  wave_barrier(); /* the obs row overlays u: finish reading it before a second pass reuses the region */
  /* HeightMap that follows the base (gq_batch_set_heightmap): the sensor's rays from the NEW base pose and heading, cast by the env's own
   * wavefront here instead of by a kernel of their own behind every step (config 5: 9.4 us + a launch boundary per step).  World-geom variants
   * only; on a flat scene the host keeps launching heightmap_kernel (a ray hits the floor: nothing to fuse). */
#ifndef GQ_NO_FUSED_HM /* (development builds: A/B of the epilogue's cost in the variants that carry it) */
  if constexpr (BOXES) if (a.heightmap) { /* wave-uniform */
    const GQ_MODEL GqDevBatch& B = Bt;
    float syaw, cyaw; /* heading as scipy's as_euler('xyz')[2] of the new orientation takes it (S11): atan2(R10, R00), at the gimbal pole atan2(-R01, R11) */
    if (fabsf(Rn[6]) < 0.9999999f) { syaw = Rn[3]; cyaw = Rn[0]; } else { syaw = -Rn[1]; cyaw = Rn[4]; }
    const float hyp2 = syaw * syaw + cyaw * cyaw, inv = hyp2 > 0.0f ? fast_rsqrt(hyp2) : 0.0f;
    syaw *= inv; cyaw = hyp2 > 0.0f ? cyaw * inv : 1.0f;
    heightmap_rays(m, bxn_d, byn_d, (double)znew, cyaw, syaw, B.hm_rows, B.hm_cols, B.hm_dx, B.hm_dy, a.heightmap + (size_t)env * (size_t)(B.hm_rows * B.hm_cols) * 3);
  }
#endif
  /* in-episode resampling (quadruped_env.py:292-305): the user's step only; a redraw acts from the next step on */
  if (pass == 0 && e_h9) resample_wave<PUB>(a, env);
  GQ_SUB(W, 2, 12); /* resampling */
  GQ_TICK(13);
  if constexpr (DBG) if (timing && lane == 0) call.debug[(size_t)env * GQ_DBG_SIZE + GQ_DBG_TIMER + 25] = (float)(wall_clock64() & 0xFFFFF);
#undef GQ_TICK
  return terminated;
}


/* ------------------------------------------------------------------ reset: state write + lift loop
 * QuadrupedEnv.reset (quadruped_env.py:332-395) for one env.  Random draws come from a counter-based generator
 * (Philox4x32-10, key = seed, counter = (draw/4, episode, env, 0x5eed)); the reference uses numpy's global
 * MT19937, which a batch cannot reproduce - draw ORDER and distributions are kept, the stream is documented in
 * DESIGN.md and restated in tests/philox_ref.py.  On the flat floor a pure z shift moves every distance by the
 * same amount, so the lift loop runs on the distances of one kinematics pass. */
struct ResetCfgDev {
  uint32_t seed_lo, seed_hi;
  int32_t random;
  float q_pos_amp, q_vel_amp, roll_sweep, pitch_sweep, hip_height;
  float lin_vel_range[2], ang_vel_range[2], friction_range[2];
  int32_t cmd_forward, cmd_random, cmd_rotate, cmd_human, env_id_offset;
  int32_t cmd_reset;      /* 'reset' in base_vel_command_type: reset restarts the redraw interval (gq_batch_set_resampling) */
};
struct ResetArgs {
  const GqDevModel* model;
  const float* vx; const float* vy; const float* vz;
  const uint8_t* mask; const double* qpos_new; const float* qvel_new;
  double* qpos; float* qvel; float* qacc; float* warm; float* applied; float* time; float* cmd; float* friction_next;
  int32_t* step_num; int32_t* episode;
  int32_t* h9;            /* resampling counters (see StepArgs), may be NULL */
  uint8_t* lift_pending;  /* library scratch [N]: 1 = the env's next first-pass step performs the lift (explicit gq_reset: the
                           * reset kernel and the reset's mj_step are two launches); NULL inside a fused auto-reset */
  uint8_t* lift_failed;
  uint8_t* clear_terminated; uint8_t* clear_truncated; uint8_t* clear_invalid; /* explicit reset(): flags zeroed; NULL inside a fused auto-reset */
  ResetCfgDev cfg;
};

/* draw indices */
enum { RN_QPOS = 0, RN_QVEL = 12, RN_X = 24, RN_Y = 25, RN_ROLL = 26, RN_PITCH = 27, RN_VNORM = 28, RN_HEADING = 29,
       RN_YAWDOT = 30, RN_FRICTION = 31, RN_VEL_INTERVAL = 32 };

#define GQ_LIFT_RULE_ITERS 4
#ifndef GQ_LIFT_CAP
#define GQ_LIFT_CAP 100 /* iteration cap of the lift loop (the reference raises RuntimeError at 100); development builds lower it to measure what the loop costs */
#endif
template <bool BOXES, bool PRIM = true, bool PUB = false>
__device__ __forceinline__ int reset_wave(const ResetArgs& a, WaveMem& W, const int env0 = 0) {
  int lane_o = lane_id(), env_o = wave_index() + uniform(env0);
  opaque(lane_o); opaque_s(env_o); /* see step_wave: no address arithmetic may be hoisted to the kernel prologue */
  const int lane = lane_o, env = env_o;
  const GQ_MODEL GqDevModel& m = *mptr(a.model);
  const ResetCfgDev& c = a.cfg;
  const int episode = a.episode ? ldv<PUB>(a.episode + env) : 0;
  /* one uniform in [0,1) per lane < 36 (draw table: RN_*) */
  float u = 0.0f;
  if (lane < 36) {
    uint32_t x = philox4x32((uint32_t)(lane >> 2), (uint32_t)episode, (uint32_t)(env + c.env_id_offset), 0x5eedu, c.seed_lo, c.seed_hi, lane & 3);
    u = (float)(x >> 8) * (1.0f / 16777216.0f);
  }
  W.u.obs[lane] = u; /* scratch: publish the draws */
  wave_barrier();
  /* draws needed after the kinematics pass (which reuses the union) */
  const float u_vnorm = W.u.obs[RN_VNORM], u_heading = W.u.obs[RN_HEADING], u_yawdot = W.u.obs[RN_YAWDOT], u_fric = W.u.obs[RN_FRICTION];
  const float u_interval = W.u.obs[RN_VEL_INTERVAL];
  const bool explicit_state = a.qpos_new != nullptr;
  double q = 0.0;
  double spawn_x = (double)m.key_qpos[0], spawn_y = (double)m.key_qpos[1]; /* wave-uniform: base x/y of the lift loop (BOXES) */
  float qv = 0.0f;
  if (explicit_state) {
    if (lane < 19) q = gptr(a.qpos_new)[(size_t)env * 19 + lane];
    if (lane < 18) qv = gptr(a.qvel_new)[(size_t)env * 18 + lane];
  } else {
    if (lane < 19) q = (double)m.key_qpos[lane];
    if (c.random) {
      if (lane >= 7 && lane < 19) q += (double)((2.0f * W.u.obs[RN_QPOS + lane - 7] - 1.0f) * c.q_pos_amp);
      if (lane >= 6 && lane < 18) qv = (2.0f * W.u.obs[RN_QVEL + lane - 6] - 1.0f) * c.q_vel_amp;
      /* xy ~ U(terrain_limits[0], [1]) x U([2], [3]) (np.random.uniform(low, high) = low + (high-low) u) */
      const double x = m.terrain_limits[0] + (m.terrain_limits[1] - m.terrain_limits[0]) * (double)W.u.obs[RN_X];
      const double y = m.terrain_limits[2] + (m.terrain_limits[3] - m.terrain_limits[2]) * (double)W.u.obs[RN_Y];
      const float roll = (2.0f * W.u.obs[RN_ROLL] - 1.0f) * c.roll_sweep, pitch = (2.0f * W.u.obs[RN_PITCH] - 1.0f) * c.pitch_sweep;
      /* heading towards the origin (math_utils.py:37-51); fp32 atan2f on purpose: the f64 routine's constant table
       * was being hoisted to the kernel prologue and spilled by every wave of every step */
      const float yaw = atan2_fast((float)(-y), (float)(-x));
      float cr, sr, cp, sp, cy, sy;
      sincos_small(0.5f * roll, sr, cr); sincos_small(0.5f * pitch, sp, cp); sincos_small(0.5f * yaw, sy, cy);
      spawn_x = x; spawn_y = y;
      if (lane == 0) q = x;
      if (lane == 1) q = y;
      if (lane == 2) q = (double)c.hip_height;
      /* Rotation.from_euler('xyz', [roll, pitch, yaw]).as_quat(scalar_first=True) */
      if (lane == 3) q = (double)(cr * cp * cy + sr * sp * sy);
      if (lane == 4) q = (double)(sr * cp * cy - cr * sp * sy);
      if (lane == 5) q = (double)(cr * sp * cy + sr * cp * sy);
      if (lane == 6) q = (double)(cr * cp * sy - sr * sp * cy);
    }
  }
  if (lane == 2) W.basez = (float)q;
  else if (lane >= 3 && lane < 7) W.qb[lane - 3] = (float)q;
  else if (lane >= 7 && lane < 19) W.qj[lane - 7] = (float)q;
  wave_barrier();
  float dz = 0.0f;
  int failed = 0;
#ifdef GQ_RESET_STAMPS /* development probe (tools/reset_probe.py): cycle stamps of this function, left in the env's qfrc_applied row */
  const long long rs_t0 = cycles(); long long rs_t1 = rs_t0, rs_t2 = rs_t0; int rs_iters = 0, rs_scans = 0, rs_cand = 0;
#endif
  /* scenes without world boxes / height field: the lift loop runs inside the reset's own mj_step (step_wave, S6) on that
   * step's kinematics and collision scan - this function only writes the spawn state and says that a lift is due */
  const bool flat_scene = !BOXES || (m.nbox == 0 && m.hf_nrow == 0); /* BOXES variants also serve flat scenes with robot self-collision */
  const int lift_due = (flat_scene && !explicit_state) ? 1 : 0;
  if (!flat_scene && !explicit_state) {
    stage_kinematics(W, link_fetch(m, lane));
    wave_barrier();
    stage_collision_scan(W, m, mptr(a.vx), mptr(a.vy), mptr(a.vz), true, foot_fetch(m, lane));
    /* distances and margins of everything attached to a calf body (feet_contact_state is body-level) */
    /* floor: lane = collision item (feet 0-3, then the link geoms), every candidate point of the plane narrow phase */
    FloorCand FC;
    FC.n = 0; FC.key = 0xE4;
    float margin = 0.0f;
    if (lane < 4 + m.nlg) {
      const bool calf_item = lane < 4 || (m.lg[lane - 4].body > 0 && (m.lg[lane - 4].body - 1) % 3 == 2);
      if (calf_item) { /* item records are in contact order: look the lane's item up by its code */
        int it = 0;
        for (int q = 0; q < 4 + m.nlg; q++) it = m.item[q].code == lane ? q : it;
        const ItemRegs IR = item_fetch(m, it);
        floor_candidates(W, IR, FC); margin = IR.margin;
      }
    }
    auto floor_pen = [&](const float dzz) { /* largest |dist| among the item's floor contacts at lift dzz (0: none) */
      float pen = 0.0f;
#pragma unroll
      for (int k = 0; k < 4; k++) pen = (k < FC.n && FC.dist[k] + dzz < margin) ? fmaxf(pen, fabsf(FC.dist[k] + dzz)) : pen;
      return pen;
    };
    if constexpr (!BOXES) {
      for (int it = 0; it < 100; it++) {
        const float pen = wave_max(floor_pen(dz));
        if (!(pen > 0.0f)) break;
        dz += 1.1f * pen;
      }
      failed = wave_max(floor_pen(dz)) > 0.0f;
    } else {
      /* with world boxes a lift changes the box distances unevenly: every iteration re-evaluates the calf-body items against
       * the floor (shifted) and against the boxes near the lifted robot, like the reference's mj_step1 per iteration */
      /* The reference's rule (z += 1.1 max|dist| until nothing touches) converges geometrically when a foot leaves a box
       * through a steep side face (gain 1.1 |dist| n_z per iteration) and then raises RuntimeError after 100 iterations.
       * Here the rule is followed for GQ_LIFT_RULE_ITERS iterations; after that the lift is whatever clears the top of
       * every box still touched (boxes are convex: the pose above them is free) - a handful of scans instead of 100. */
      V3 calf_c; float calf_r;
      item_sphere(W, m, true, calf_c, calf_r);
#ifdef GQ_RESET_STAMPS
      rs_t1 = cycles();
#endif
      const PrimLane PLL = prim_lane(W, m, item_fetch(m, lane < 4 + m.nlg ? lane : 0), PRIM && lane < 4 + m.nlg, PRIM); /* lane = position in con_order, as box_item_scan expects */
      { item_obb_store(W, m); wave_barrier(); } /* (the spawn pose's boxes; a lift moves the world boxes down - zoff - not the robot) */
      for (int it = 0; it <= GQ_LIFT_CAP; it++) {
        float pen = floor_pen(dz);
        float clear = 0.0f; /* lift that takes the touching item above the box altogether */
        uint64_t cand[2];
        box_candidates(W, m, spawn_x, spawn_y, dz, cand, calf_c, calf_r); /* around the lifted base; calf items only */
#ifdef GQ_RESET_STAMPS
        rs_iters++; rs_cand = imax(rs_cand, popc64(cand[0]) + popc64(cand[1])); rs_scans += popc64(cand[0]) + popc64(cand[1]);
#endif
        for (int half = 0; half < 2; half++) {
          uint64_t todo = cand[half];
          while (todo) { /* wave-uniform */
            const int b = half * GQ_WAVE + ffs64(todo);
            todo &= todo - 1;
            PairHit BH;
            if (!box_item_scan<PRIM, 1, true>(W, m, mptr(a.vx), mptr(a.vy), mptr(a.vz), b, spawn_x, spawn_y, dz, calf_c, calf_r, PLL, BH)) continue;
            if (lane < 4 + m.nlg) {
              const int code = m.con_order[lane];
              const bool calf = code < 4 || (m.lg[code - 4].body > 0 && (m.lg[code - 4].body - 1) % 3 == 2);
#pragma unroll
              for (int k = 0; k < (PRIM ? 4 : 1); k++)
                if (calf && k < BH.n && BH.dist[k] < m.boxmix[m.box[b].cls][code].margin) {
                  const float bd = BH.dist[k];
                  pen = fmaxf(pen, fabsf(bd));
                  const GQ_MODEL GqDevBox& B = m.box[b];
                  const float ztop = B.pos[2] + fabsf(B.mat[6]) * B.size[0] + fabsf(B.mat[7]) * B.size[1] + fabsf(B.mat[8]) * B.size[2];
                  /* contact point is midway between the surfaces: the item's lowest point is at most |bd| + its radius below */
                  clear = fmaxf(clear, ztop - (BH.pos[k].z + dz) + fabsf(bd) + 0.02f);
                }
            }
            wave_barrier();
          }
        }
        if (m.hf_nrow > 0) { /* the height field: a lift by dz changes a distance by dz * n_z */
          float hd; V3 hn, hpt;
          if (hfield_item_scan(W, m, mptr(a.vx), mptr(a.vy), mptr(a.vz), spawn_x, spawn_y, dz, calf_c, calf_r, hd, hn, hpt)) {
            if (lane < 4 + m.nlg) {
              const int code = m.con_order[lane];
              const bool calf = code < 4 || (m.lg[code - 4].body > 0 && (m.lg[code - 4].body - 1) % 3 == 2);
              if (calf && hd < m.boxmix[m.hf_cls][code].margin) {
                pen = fmaxf(pen, fabsf(hd));
                clear = fmaxf(clear, fabsf(hd) / fmaxf(hn.z, 0.2f) + 0.005f);
              }
            }
            wave_barrier();
          }
        }
        pen = wave_max(pen);
        failed = pen > 0.0f;
        if (!failed || it == GQ_LIFT_CAP) break;
#ifdef GQ_EMU_TRACE
        if (lane == 0 && getenv("GQ_EMU_TRACE")) printf("lift it %d dz %.4f pen %.5f cand %d\n", it, (double)dz, (double)pen, popc64(cand[0]) + popc64(cand[1]));
#endif
        dz += it < GQ_LIFT_RULE_ITERS ? 1.1f * pen : fmaxf(1.1f * pen, wave_max(clear));
      }
    }
  }
#ifdef GQ_RESET_STAMPS
  rs_t2 = cycles();
#endif
  if (lane < 19) gptr(a.qpos)[(size_t)env * 19 + lane] = lane == 2 ? q + (double)dz : q;
  if (lane < 18) {
    gptr(a.qvel)[(size_t)env * 18 + lane] = qv;
    gptr(a.qacc)[(size_t)env * 18 + lane] = 0.0f;
    gptr(a.warm)[(size_t)env * 18 + lane] = 0.0f;
    if (a.applied) gptr(a.applied)[(size_t)env * 18 + lane] = 0.0f;
  }
#ifdef GQ_RESET_STAMPS
  if (lane == 0 && a.applied) { GQ_GLOBAL float* S = gptr(a.applied) + (size_t)env * 18 + 6;
    S[0] = 1e-9f * (float)(rs_t1 - rs_t0); S[1] = 1e-9f * (float)(rs_t2 - rs_t1); S[2] = 1e-9f * (float)rs_iters; S[3] = 1e-9f * (float)rs_scans; S[4] = 1e-9f * (float)rs_cand; }
#endif
  if (lane == 0) {
    gptr(a.time)[env] = 0.0f;
    gptr(a.step_num)[env] = -1; /* the reset's own mj_step brings it to 0 (:332, :397) */
    if (a.episode) gptr(a.episode)[env] = episode + 1;
    if (a.lift_failed && !lift_due) gptr(a.lift_failed)[env] = (uint8_t)failed;
    if (a.lift_pending) gptr(a.lift_pending)[env] = (uint8_t)lift_due;
    if (a.clear_terminated) { gptr(a.clear_terminated)[env] = 0; gptr(a.clear_truncated)[env] = 0; gptr(a.clear_invalid)[env] = 0; }
    /* _sample_ref_vel (:1046-1072) */
    if (a.cmd) {
      float norm = 0.0f, heading = 0.0f, yaw_dot = 0.0f;
      if (c.cmd_forward) norm = c.lin_vel_range[0] + (c.lin_vel_range[1] - c.lin_vel_range[0]) * u_vnorm;
      else if (c.cmd_random) {
        norm = c.lin_vel_range[0] + (c.lin_vel_range[1] - c.lin_vel_range[0]) * u_vnorm;
        heading = (2.0f * u_heading - 1.0f) * 3.14159265358979f;
      }
      if (c.cmd_rotate) yaw_dot = c.ang_vel_range[0] + (c.ang_vel_range[1] - c.ang_vel_range[0]) * u_yawdot;
      float sh, ch;
      sincos_small(heading, sh, ch);
      gptr(a.cmd)[(size_t)env * 4 + 0] = norm * ch; gptr(a.cmd)[(size_t)env * 4 + 1] = norm * sh;
      gptr(a.cmd)[(size_t)env * 4 + 2] = 0.0f; gptr(a.cmd)[(size_t)env * 4 + 3] = yaw_dot;
      /* 'reset' command types restart their redraw interval (:1068-1070) */
      if (a.h9 && c.cmd_reset) { gptr(a.h9)[(size_t)env * 6 + 0] = 0; gptr(a.h9)[(size_t)env * 6 + 1] = 1000 + (int)(2000.0f * u_interval); }
    }
    if (a.friction_next)
      gptr(a.friction_next)[env] = c.friction_range[0] + (c.friction_range[1] - c.friction_range[0]) * u_fric;
  }
  GQ_LIFT_DUE(W) = lift_due; /* (every lane: the same word) step_wave's S6 reads it */
  return lift_due;
}


struct FusedArgs {      /* device-resident argument block of step_kernel */
  StepArgs s;
  ResetArgs r;          /* used when StepCall.auto_reset != 0 */
};

}  // namespace gq
