/*
 * gq_oracle.c - CPU fp64 single-env restatement of the physics step on the
 * QuadrupedEnv.step() hot path.  TEST INFRASTRUCTURE ONLY: nothing in the
 * product path (gym_quadruped_amd/) may link, load or call this file; it is
 * used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 *
 * PARITY UNPINNED for the mj_step part: the reference delegates the whole
 * step to the third-party dependency `mujoco` (pyproject.toml:30,
 * "mujoco>=3.10.0", unpinned; call site gym_quadruped/quadruped_env.py:271),
 * which is neither vendored under the reference tree nor installable here, and
 * the reference's own test (tests/env_test.py:35-46) pins shapes only.  This
 * file therefore restates MuJoCo's published algorithm (documentation chapter
 * "Computation" + the open-source engine's function structure), function by
 * function:
 *
 *   mj_kinematics / mj_comPos          -> gqo_kinematics, gqo_com_pos
 *   mj_crb / mj_factorM / mj_solveM    -> gqo_crb, gqo_factor_m, gqo_solve_m
 *   mj_comVel / mj_rne / mj_passive    -> gqo_com_vel, gqo_rne, passive in gqo_fwd_velocity
 *   mj_fwdActuation                    -> gqo_fwd_actuation
 *   mj_collision (plane vs robot geom) -> gqo_collision
 *   mj_makeConstraint + mj_makeImpedance + mj_diagApprox -> gqo_make_constraint
 *   mj_solNewton / mj_solPGS           -> gqo_sol_newton, gqo_sol_pgs
 *   mj_Euler / mj_integratePos         -> gqo_euler
 *   mj_jac, mj_contactForce, mj_fullM  -> gqo_jac, gqo_contact_force, M[][]
 *
 * It is pinned instead by physical invariants (tests/test_oracle_invariants.py)
 * and, for the observation layer above it, by golden vectors captured from the
 * importable numpy part of the reference (tests/golden/).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gq.h"
#include "gq_convex.h"

#define NQ 19
#define NV 18
#define NU 12
#define NB 14
#define NJ 13
#define NG 96
#define NCON 160
#define NEFC 1024
#define MINVAL 1e-15
#define MINIMP 0.0001
#define MINMU 1e-5
#define MAXIMP 0.9999

enum { EFC_FRICTION_DOF = 0, EFC_LIMIT_JOINT = 1, EFC_CONTACT_FRICTIONLESS = 2, EFC_CONTACT_PYRAMIDAL = 3,
       EFC_CONTACT_ELLIPTIC = 4 };

typedef struct {
  double dist, pos[3], frame[9], includemargin, friction[5], solref[2], solimp[5], mu;
  double tiegap; /* depth gap to the second deepest vertex of the geom's cloud (test diagnostics) */
  int geom, body, dim, efc_address;
  int geom1, body1; /* the other geom: -1 / 0 for a world geom (floor, box, height field), else a robot geom and its body */
} Contact;

typedef struct GqOracle {
  GqModelDesc d;
  void* owned[64];
  int nowned;
  /* state (mjData) */
  double qpos[NQ], qvel[NV], qacc[NV], qacc_warmstart[NV], ctrl[NU], qfrc_applied[NV], time;
  double friction; /* <0: XML frictions; >=0: floor+feet tangential coefficient (quadruped_env.py:1277-1298) */
  /* position stage */
  double xpos[NB][3], xquat[NB][4], xmat[NB][9], xipos[NB][3], ximat[NB][9];
  double xanchor[NJ][3], xaxis[NJ][3], subtree_com[NB][3];
  double cinert[NB][10], crb[NB][10], cdof[NV][6];
  double M[NV][NV], LD[NV][NV];
  double geom_xpos[NG][3], geom_xmat[NG][9];
  /* velocity stage */
  double cvel[NB][6], cdof_dot[NV][6], cacc[NB][6], cfrc[NB][6];
  double qfrc_bias[NV], qfrc_passive[NV], qfrc_actuator[NV], qfrc_smooth[NV], qacc_smooth[NV], qfrc_constraint[NV];
  /* contacts + constraints */
  int ncon, nefc;
  Contact contact[NCON];
  int efc_type[NEFC], efc_id[NEFC];
  double efc_J[NEFC][NV], efc_pos[NEFC], efc_margin[NEFC], efc_frictionloss[NEFC], efc_diagApprox[NEFC];
  double efc_R[NEFC], efc_D[NEFC], efc_vel[NEFC], efc_aref[NEFC], efc_b[NEFC], efc_force[NEFC];
  double efc_solref[NEFC][2], efc_solimp[NEFC][5];
  int solver_niter;
  /* IMU site on the base body + last mj_sensorAcc / mj_sensorVel readings (noise-free) */
  double imu_pos[3], imu_quat[4], imu_acc[3], imu_gyro[3];
  int warning; /* bad qpos/qvel/qacc seen (mj_checkPos/Vel/Acc) */
  double cloud_aabb[64][6]; int cloud_aabb_ok[64]; /* geom-frame box of every cloud (centre, half extents): mid phase of the convex pairs */
} GqOracle;

/* ------------------------------------------------------------------ small vector helpers */
static inline double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void cross3(double* r, const double* a, const double* b) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
static inline void mulmatvec3(double* r, const double* m, const double* v) {
  double x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2], y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2],
         z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static inline void mulmatTvec3(double* r, const double* m, const double* v) {
  double x = m[0] * v[0] + m[3] * v[1] + m[6] * v[2], y = m[1] * v[0] + m[4] * v[1] + m[7] * v[2],
         z = m[2] * v[0] + m[5] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static void mulmat3(double* r, const double* a, const double* b) {
  double t[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) t[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
  memcpy(r, t, sizeof t);
}
static void quat_mul(double* r, const double* a, const double* b) {
  double t[4] = {a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                 a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]};
  memcpy(r, t, sizeof t);
}
static void quat_normalize(double* q) {
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
  for (int i = 0; i < 4; i++) q[i] /= n;
}
static void quat2mat(double* m, const double* q) {
  double w = q[0], x = q[1], y = q[2], z = q[3];
  m[0] = w * w + x * x - y * y - z * z; m[4] = w * w - x * x + y * y - z * z; m[8] = w * w - x * x - y * y + z * z;
  m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y); m[3] = 2 * (x * y + w * z);
  m[5] = 2 * (y * z - w * x); m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x);
}
static void axis_angle2quat(double* q, const double* axis, double angle) {
  double s = sin(0.5 * angle);
  q[0] = cos(0.5 * angle); q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s;
}
/* spatial motion cross product  res = vel x_m v   (mju_crossMotion) */
static void cross_motion(double* res, const double* vel, const double* v) {
  double a[3], b[3];
  cross3(res, vel, v);
  cross3(a, vel, v + 3); cross3(b, vel + 3, v);
  res[3] = a[0] + b[0]; res[4] = a[1] + b[1]; res[5] = a[2] + b[2];
}
/* spatial force cross product  res = vel x_f f   (mju_crossForce) */
static void cross_force(double* res, const double* vel, const double* f) {
  double a[3], b[3];
  cross3(a, vel, f); cross3(b, vel + 3, f + 3);
  res[0] = a[0] + b[0]; res[1] = a[1] + b[1]; res[2] = a[2] + b[2];
  cross3(res + 3, vel, f + 3);
}
/* res = I * v, I = [Ixx Iyy Izz Ixy Ixz Iyz | m*c (3) | m]   (mju_mulInertVec) */
static void mul_inert_vec(double* res, const double* i, const double* v) {
  res[0] = i[0] * v[0] + i[3] * v[1] + i[4] * v[2] - i[8] * v[4] + i[7] * v[5];
  res[1] = i[3] * v[0] + i[1] * v[1] + i[5] * v[2] + i[8] * v[3] - i[6] * v[5];
  res[2] = i[4] * v[0] + i[5] * v[1] + i[2] * v[2] - i[7] * v[3] + i[6] * v[4];
  res[3] = i[8] * v[1] - i[7] * v[2] + i[9] * v[3];
  res[4] = i[6] * v[2] - i[8] * v[0] + i[9] * v[4];
  res[5] = i[7] * v[0] - i[6] * v[1] + i[9] * v[5];
}

/* ------------------------------------------------------------------ model copy */
static void* own(GqOracle* o, const void* src, size_t bytes) {
  if (!src || !bytes) return NULL;
  void* p = malloc(bytes);
  memcpy(p, src, bytes);
  o->owned[o->nowned++] = p;
  return p;
}
#define OWN(field, count) o->d.field = own(o, desc->field, sizeof(*desc->field) * (size_t)(count))

static char g_err[256];
const char* gqo_last_error(void) { return g_err; }

int gqo_create(const GqModelDesc* desc, GqOracle** out) {
  if (desc->struct_size != (int32_t)sizeof(GqModelDesc)) {
    snprintf(g_err, sizeof g_err, "oracle: GqModelDesc.struct_size %d, expected %d (include/gq.h ABI %d)", desc->struct_size, (int)sizeof(GqModelDesc), GQ_ABI_VERSION);
    return GQ_EINVAL;
  }
  if (desc->nq != NQ || desc->nv != NV || desc->nu > NU || desc->nbody > NB || desc->njnt > NJ || desc->ngeom > NG) {
    snprintf(g_err, sizeof g_err, "oracle: unsupported sizes nq=%d nv=%d nu=%d nbody=%d ngeom=%d", desc->nq, desc->nv,
             desc->nu, desc->nbody, desc->ngeom);
    return GQ_EINVAL;
  }
  GqOracle* o = calloc(1, sizeof(GqOracle));
  if (!o) return GQ_ENOMEM;
  o->d = *desc;
  int nb = desc->nbody, nj = desc->njnt, nv = desc->nv, ng = desc->ngeom, nu = desc->nu;
  OWN(body_parentid, nb); OWN(body_pos, 3 * nb); OWN(body_quat, 4 * nb); OWN(body_ipos, 3 * nb); OWN(body_iquat, 4 * nb);
  OWN(body_mass, nb); OWN(body_inertia, 3 * nb); OWN(body_jntadr, nb); OWN(body_jntnum, nb); OWN(body_invweight0, 2 * nb);
  OWN(jnt_type, nj); OWN(jnt_bodyid, nj); OWN(jnt_qposadr, nj); OWN(jnt_dofadr, nj); OWN(jnt_pos, 3 * nj); OWN(jnt_axis, 3 * nj);
  OWN(jnt_limited, nj); OWN(jnt_range, 2 * nj); OWN(jnt_margin, nj); OWN(jnt_solref, 2 * nj); OWN(jnt_solimp, 5 * nj);
  OWN(jnt_actfrclimited, nj); OWN(jnt_actfrcrange, 2 * nj); OWN(qpos0, desc->nq);
  OWN(dof_bodyid, nv); OWN(dof_jntid, nv); OWN(dof_parentid, nv); OWN(dof_damping, nv); OWN(dof_armature, nv);
  OWN(dof_frictionloss, nv); OWN(dof_solref, 2 * nv); OWN(dof_solimp, 5 * nv); OWN(dof_invweight0, nv);
  OWN(geom_bodyid, ng); OWN(geom_pos, 3 * ng); OWN(geom_quat, 4 * ng); OWN(geom_cloudid, ng); OWN(geom_friction, 3 * ng);
  OWN(geom_margin, ng); OWN(geom_gap, ng); OWN(geom_condim, ng); OWN(geom_priority, ng); OWN(geom_solref, 2 * ng);
  OWN(geom_solimp, 5 * ng); OWN(geom_solmix, ng); OWN(geom_rbound, ng);
  OWN(cloud_vertadr, desc->ncloud); OWN(cloud_vertnum, desc->ncloud); OWN(cloud_radius, desc->ncloud);
  OWN(vert_pos, 3 * desc->nvert);
  OWN(actuator_trnid, nu); OWN(actuator_gear, nu); OWN(actuator_ctrllimited, nu); OWN(actuator_ctrlrange, 2 * nu);
  OWN(actuator_forcelimited, nu); OWN(actuator_forcerange, 2 * nu);
  memcpy(o->qpos, o->d.qpos0, sizeof(double) * NQ);
  o->friction = -1.0;
  o->imu_quat[0] = 1.0;
  *out = o;
  return GQ_OK;
}

void gqo_destroy(GqOracle* o) {
  if (!o) return;
  for (int i = 0; i < o->nowned; i++) free(o->owned[i]);
  free(o);
}

/* ------------------------------------------------------------------ mj_kinematics */
static void gqo_kinematics(GqOracle* o) {
  const GqModelDesc* m = &o->d;
  memset(o->xpos[0], 0, sizeof o->xpos[0]);
  o->xquat[0][0] = 1; o->xquat[0][1] = o->xquat[0][2] = o->xquat[0][3] = 0;
  quat2mat(o->xmat[0], o->xquat[0]);
  memset(o->xipos[0], 0, sizeof o->xipos[0]);
  quat2mat(o->ximat[0], o->xquat[0]);
  for (int b = 1; b < m->nbody; b++) {
    int p = m->body_parentid[b];
    double pos[3], quat[4], tmp[3];
    mulmatvec3(tmp, o->xmat[p], m->body_pos + 3 * b);
    for (int k = 0; k < 3; k++) pos[k] = o->xpos[p][k] + tmp[k];
    quat_mul(quat, o->xquat[p], m->body_quat + 4 * b);
    for (int j = m->body_jntadr[b]; j >= 0 && j < m->body_jntadr[b] + m->body_jntnum[b]; j++) {
      int qa = m->jnt_qposadr[j];
      if (m->jnt_type[j] == 0) { /* free: pose straight from qpos, quaternion normalised */
        memcpy(pos, o->qpos + qa, sizeof pos);
        memcpy(quat, o->qpos + qa + 3, sizeof quat);
        quat_normalize(quat);
        memcpy(o->xanchor[j], pos, sizeof pos);
        o->xaxis[j][0] = 0; o->xaxis[j][1] = 0; o->xaxis[j][2] = 1;
      } else { /* hinge: anchor/axis in the frame BEFORE this joint's rotation, angle relative to qpos0 */
        double R[9], ql[4], a[3];
        quat2mat(R, quat);
        mulmatvec3(a, R, m->jnt_pos + 3 * j);
        for (int k = 0; k < 3; k++) o->xanchor[j][k] = pos[k] + a[k];
        mulmatvec3(o->xaxis[j], R, m->jnt_axis + 3 * j);
        axis_angle2quat(ql, m->jnt_axis + 3 * j, o->qpos[qa] - m->qpos0[qa]);
        quat_mul(quat, quat, ql);
        /* off-centre rotation correction: keep the anchor fixed */
        quat2mat(R, quat);
        mulmatvec3(a, R, m->jnt_pos + 3 * j);
        for (int k = 0; k < 3; k++) pos[k] = o->xanchor[j][k] - a[k];
      }
    }
    quat_normalize(quat);
    memcpy(o->xpos[b], pos, sizeof pos);
    memcpy(o->xquat[b], quat, sizeof quat);
    quat2mat(o->xmat[b], quat);
    mulmatvec3(tmp, o->xmat[b], m->body_ipos + 3 * b);
    for (int k = 0; k < 3; k++) o->xipos[b][k] = pos[k] + tmp[k];
    double qi[4];
    quat_mul(qi, quat, m->body_iquat + 4 * b);
    quat2mat(o->ximat[b], qi);
  }
  for (int g = 0; g < m->ngeom; g++) {
    int b = m->geom_bodyid[g];
    double tmp[3], q[4];
    mulmatvec3(tmp, o->xmat[b], m->geom_pos + 3 * g);
    for (int k = 0; k < 3; k++) o->geom_xpos[g][k] = o->xpos[b][k] + tmp[k];
    quat_mul(q, o->xquat[b], m->geom_quat + 4 * g);
    quat2mat(o->geom_xmat[g], q);
  }
}

/* ------------------------------------------------------------------ mj_comPos */
static void gqo_com_pos(GqOracle* o) {
  const GqModelDesc* m = &o->d;
  double mass[NB];
  for (int b = 0; b < m->nbody; b++) {
    mass[b] = m->body_mass[b];
    for (int k = 0; k < 3; k++) o->subtree_com[b][k] = m->body_mass[b] * o->xipos[b][k];
  }
  for (int b = m->nbody - 1; b > 0; b--) {
    int p = m->body_parentid[b];
    mass[p] += mass[b];
    for (int k = 0; k < 3; k++) o->subtree_com[p][k] += o->subtree_com[b][k];
  }
  for (int b = 0; b < m->nbody; b++)
    for (int k = 0; k < 3; k++) o->subtree_com[b][k] = mass[b] > MINVAL ? o->subtree_com[b][k] / mass[b] : o->xipos[b][k];
  /* every robot body has root body 1 (the floating base): com-based frame origin = subtree_com[1] */
  const double* O = o->subtree_com[1];
  memset(o->cinert[0], 0, sizeof o->cinert[0]);
  for (int b = 1; b < m->nbody; b++) { /* mju_inertCom */
    double d[3], tmp[9], I[9], mb = m->body_mass[b];
    for (int k = 0; k < 3; k++) d[k] = o->xipos[b][k] - O[k];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) tmp[3 * i + j] = o->ximat[b][3 * i + j] * m->body_inertia[3 * b + j];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += tmp[3 * i + k] * o->ximat[b][3 * j + k];
        I[3 * i + j] = s;
      }
    double dd = dot3(d, d);
    double* c = o->cinert[b];
    c[0] = I[0] + mb * (dd - d[0] * d[0]); c[1] = I[4] + mb * (dd - d[1] * d[1]); c[2] = I[8] + mb * (dd - d[2] * d[2]);
    c[3] = I[1] - mb * d[0] * d[1]; c[4] = I[2] - mb * d[0] * d[2]; c[5] = I[5] - mb * d[1] * d[2];
    c[6] = mb * d[0]; c[7] = mb * d[1]; c[8] = mb * d[2]; c[9] = mb;
  }
  for (int j = 0; j < m->njnt; j++) { /* cdof = [axis ; axis x (O - anchor)]  (mju_dofCom) */
    int b = m->jnt_bodyid[j], d = m->jnt_dofadr[j];
    double off[3];
    for (int k = 0; k < 3; k++) off[k] = O[k] - o->xanchor[j][k];
    if (m->jnt_type[j] == 0) {
      for (int k = 0; k < 3; k++) {
        memset(o->cdof[d + k], 0, sizeof o->cdof[0]);
        o->cdof[d + k][3 + k] = 1;
        double ax[3] = {o->xmat[b][k], o->xmat[b][3 + k], o->xmat[b][6 + k]};
        memcpy(o->cdof[d + 3 + k], ax, sizeof ax);
        cross3(o->cdof[d + 3 + k] + 3, ax, off);
      }
    } else {
      memcpy(o->cdof[d], o->xaxis[j], sizeof o->xaxis[j]);
      cross3(o->cdof[d] + 3, o->xaxis[j], off);
    }
  }
}

/* ------------------------------------------------------------------ mj_crb + mj_factorM */
static void gqo_crb(GqOracle* o) {
  const GqModelDesc* m = &o->d;
  memcpy(o->crb, o->cinert, sizeof o->crb);
  for (int b = m->nbody - 1; b > 0; b--) {
    int p = m->body_parentid[b];
    if (p > 0)
      for (int k = 0; k < 10; k++) o->crb[p][k] += o->crb[b][k];
  }
  memset(o->M, 0, sizeof o->M);
  for (int i = 0; i < m->nv; i++) {
    double buf[6];
    mul_inert_vec(buf, o->crb[m->dof_bodyid[i]], o->cdof[i]);
    o->M[i][i] = m->dof_armature[i];
    for (int j = i; j >= 0; j = m->dof_parentid[j]) {
      double s = 0;
      for (int k = 0; k < 6; k++) s += o->cdof[j][k] * buf[k];
      if (j == i) o->M[i][i] += s; else { o->M[i][j] = s; o->M[j][i] = s; }
    }
  }
}

/* L'DL factorisation exploiting the dof tree (mj_factorI): LD holds L below the diagonal (unit diagonal implied)
 * and D on the diagonal. */
static void factor_ld(const GqModelDesc* m, double LD[NV][NV]) {
  for (int k = m->nv - 1; k >= 0; k--) {
    for (int i = m->dof_parentid[k]; i >= 0; i = m->dof_parentid[i]) {
      double tmp = LD[k][i] / LD[k][k];
      for (int j = i; j >= 0; j = m->dof_parentid[j]) LD[i][j] -= LD[k][j] * tmp;
      LD[k][i] = tmp;
    }
  }
}
static void solve_ld(const GqModelDesc* m, double LD[NV][NV], double* x) {
  for (int k = m->nv - 1; k >= 0; k--)
    for (int i = m->dof_parentid[k]; i >= 0; i = m->dof_parentid[i]) x[i] -= LD[k][i] * x[k];
  for (int k = 0; k < m->nv; k++) x[k] /= LD[k][k];
  for (int k = 0; k < m->nv; k++)
    for (int i = m->dof_parentid[k]; i >= 0; i = m->dof_parentid[i]) x[k] -= LD[k][i] * x[i];
}
static void gqo_factor_m(GqOracle* o) {
  for (int i = 0; i < NV; i++)
    for (int j = 0; j < NV; j++) o->LD[i][j] = j <= i ? o->M[i][j] : 0.0;
  factor_ld(&o->d, o->LD);
}
static void gqo_solve_m(GqOracle* o, double* x) { solve_ld(&o->d, o->LD, x); }

/* ------------------------------------------------------------------ mj_jac: translational/rotational Jacobian of a
 * world point attached to `body` */
static void gqo_jac(const GqOracle* o, double jacp[3][NV], double jacr[3][NV], const double* point, int body) {
  const GqModelDesc* m = &o->d;
  double off[3];
  for (int k = 0; k < 3; k++) off[k] = point[k] - o->subtree_com[1][k];
  if (jacp) memset(jacp, 0, sizeof(double) * 3 * NV);
  if (jacr) memset(jacr, 0, sizeof(double) * 3 * NV);
  if (body <= 0) return;
  /* last dof of the body (bodies without joints do not occur in these models) */
  int j = m->body_jntadr[body] + m->body_jntnum[body] - 1;
  int i = m->jnt_dofadr[j] + (m->jnt_type[j] == 0 ? 5 : 0);
  for (; i >= 0; i = m->dof_parentid[i]) {
    double t[3];
    cross3(t, o->cdof[i], off);
    for (int k = 0; k < 3; k++) {
      if (jacr) jacr[k][i] = o->cdof[i][k];
      if (jacp) jacp[k][i] = o->cdof[i][3 + k] + t[k];
    }
  }
}

/* ------------------------------------------------------------------ mj_collision: floor plane (z = 0) vs robot geoms,
 * MuJoCo's plane routines restated per geom type (engine_collision_primitive.c / engine_collision_convex.c):
 *   sphere    mjraw_PlaneSphere   one point, exact
 *   capsule   mjraw_PlaneCapsule  both end spheres (the +axis end first), every one within the margin; the contact frame's
 *                                 first tangent is the capsule axis made orthogonal to the normal
 *   box       mjraw_PlaneBox      the corners at or below the box centre that are within the margin, in corner order
 *                                 (x sign = bit 0, y sign = bit 1, z sign = bit 2), at most 4
 *   cylinder  mjc_PlaneCylinder   analytic: the lowest rim point of the end cap nearer the plane, the rim point under it on
 *                                 the other cap, and two more points of the near cap at +-120 degrees (at most 4)
 *   mesh      mjc_PlaneConvex     the support vertex of the hull (MuJoCo adds up to three neighbouring hull vertices inside
 *                                 the margin; not restated - documented deviation)
 * mju_makeFrame: the normal is the frame's x axis; the y axis is the one the routine supplied (capsule) or (0,1,0) /
 * (0,0,1), made orthogonal to x; z = x cross y. */
static void make_frame(double* f) { /* f[0..2] normal, f[3..5] tangent hint or zero */
  double n = sqrt(dot3(f, f));
  for (int k = 0; k < 3; k++) f[k] /= n;
  if (sqrt(dot3(f + 3, f + 3)) < 0.5) {
    f[3] = f[4] = f[5] = 0;
    if (f[1] < 0.5 && f[1] > -0.5) f[4] = 1; else f[5] = 1;
  }
  double t = dot3(f, f + 3);
  for (int k = 0; k < 3; k++) f[3 + k] -= t * f[k];
  n = sqrt(dot3(f + 3, f + 3));
  if (n < MINVAL) { f[3] = 1; f[4] = 0; f[5] = 0; } /* mju_normalize3 of a null vector */
  else for (int k = 0; k < 3; k++) f[3 + k] /= n;
  cross3(f + 6, f, f + 3);
}
static void set_frame(Contact* c, const double* normal, const double* tangent_hint) {
  memcpy(c->frame, normal, 3 * sizeof(double));
  for (int k = 0; k < 3; k++) c->frame[3 + k] = tangent_hint ? tangent_hint[k] : 0.0;
  make_frame(c->frame);
}

static int is_foot(const GqModelDesc* m, int g) {
  for (int l = 0; l < GQ_NLEG; l++)
    if (m->feet_geomid[l] == g) return 1;
  return 0;
}

/* mj_contactParam: mix the world geom (geom 1: the floor for w < 0, world box w otherwise) and robot geom g (geom 2) */
static void contact_param(const GqOracle* o, int w, int g, Contact* c) {
  const GqModelDesc* m = &o->d;
  double f1[3], f2[3];
  const int hf = w >= m->nbox; /* the height field comes after the boxes */
  memcpy(f1, w < 0 ? m->floor_friction : (hf ? m->hfield_friction : m->box_friction + 3 * w), sizeof f1);
  memcpy(f2, m->geom_friction + 3 * g, sizeof f2);
  if (o->friction >= 0) { /* _set_ground_friction (quadruped_env.py:1277-1298): geoms named ground/floor/hfield/terrain and the
                           * feet get [mu, 0.005, 0.0]; the unnamed world boxes keep their own friction (quirk B8) */
    if (w < 0) { f1[0] = o->friction; f1[1] = 0.005; f1[2] = 0.0; }
    if (is_foot(m, g)) { f2[0] = o->friction; f2[1] = 0.005; f2[2] = 0.0; }
  }
  const int w_condim = w < 0 ? m->floor_condim : (hf ? m->hfield_condim : m->box_condim[w]);
  const int w_priority = w < 0 ? m->floor_priority : (hf ? m->hfield_priority : m->box_priority[w]);
  const double w_solmix = w < 0 ? m->floor_solmix : (hf ? m->hfield_solmix : m->box_solmix[w]);
  const double w_margin = w < 0 ? m->floor_margin : (hf ? m->hfield_margin : m->box_margin[w]);
  const double w_gap = w < 0 ? m->floor_gap : (hf ? m->hfield_gap : m->box_gap[w]);
  const double* w_solref = w < 0 ? m->floor_solref : (hf ? m->hfield_solref : m->box_solref + 2 * w);
  const double* w_solimp = w < 0 ? m->floor_solimp : (hf ? m->hfield_solimp : m->box_solimp + 5 * w);
  int p1 = w_priority, p2 = m->geom_priority[g];
  double fri[3];
  if (p1 == p2) {
    c->dim = w_condim > m->geom_condim[g] ? w_condim : m->geom_condim[g];
    for (int k = 0; k < 3; k++) fri[k] = f1[k] > f2[k] ? f1[k] : f2[k];
    double s1 = w_solmix, s2 = m->geom_solmix[g], mix;
    if (s1 >= MINVAL && s2 >= MINVAL) mix = s1 / (s1 + s2);
    else if (s1 < MINVAL && s2 < MINVAL) mix = 0.5;
    else mix = s1 < MINVAL ? 0.0 : 1.0;
    const double* r1 = w_solref; const double* r2 = m->geom_solref + 2 * g;
    if (r1[0] > 0 && r2[0] > 0)
      for (int k = 0; k < 2; k++) c->solref[k] = mix * r1[k] + (1 - mix) * r2[k];
    else
      for (int k = 0; k < 2; k++) c->solref[k] = r1[k] < r2[k] ? r1[k] : r2[k];
    for (int k = 0; k < 5; k++) c->solimp[k] = mix * w_solimp[k] + (1 - mix) * m->geom_solimp[5 * g + k];
  } else {
    int floor_wins = p1 > p2;
    c->dim = floor_wins ? w_condim : m->geom_condim[g];
    memcpy(fri, floor_wins ? f1 : f2, sizeof fri);
    memcpy(c->solref, floor_wins ? w_solref : m->geom_solref + 2 * g, sizeof c->solref);
    memcpy(c->solimp, floor_wins ? w_solimp : m->geom_solimp + 5 * g, sizeof c->solimp);
  }
  c->friction[0] = c->friction[1] = fri[0]; c->friction[2] = fri[1]; c->friction[3] = c->friction[4] = fri[2];
  for (int k = 0; k < 5; k++) c->friction[k] = fmax(MINMU, c->friction[k]); /* mjMINMU */
  double margin = w_margin > m->geom_margin[g] ? w_margin : m->geom_margin[g];
  double gap = w_gap > m->geom_gap[g] ? w_gap : m->geom_gap[g];
  c->includemargin = margin - gap;
  c->mu = 0;
}


/* mj_contactParam for two ROBOT geoms g1 < g2 (self-collision) */
static void contact_param_pair(const GqOracle* o, int g1, int g2, Contact* c) {
  const GqModelDesc* m = &o->d;
  double f1[3], f2[3], fri[3];
  memcpy(f1, m->geom_friction + 3 * g1, sizeof f1); memcpy(f2, m->geom_friction + 3 * g2, sizeof f2);
  if (o->friction >= 0) { /* _set_ground_friction rewrites the feet geoms (quadruped_env.py:1277-1298) */
    if (is_foot(m, g1)) { f1[0] = o->friction; f1[1] = 0.005; f1[2] = 0.0; }
    if (is_foot(m, g2)) { f2[0] = o->friction; f2[1] = 0.005; f2[2] = 0.0; }
  }
  const int p1 = m->geom_priority[g1], p2 = m->geom_priority[g2];
  if (p1 == p2) {
    c->dim = m->geom_condim[g1] > m->geom_condim[g2] ? m->geom_condim[g1] : m->geom_condim[g2];
    for (int k = 0; k < 3; k++) fri[k] = f1[k] > f2[k] ? f1[k] : f2[k];
    double s1 = m->geom_solmix[g1], s2 = m->geom_solmix[g2], mix;
    if (s1 >= MINVAL && s2 >= MINVAL) mix = s1 / (s1 + s2);
    else if (s1 < MINVAL && s2 < MINVAL) mix = 0.5;
    else mix = s1 < MINVAL ? 0.0 : 1.0;
    const double* r1 = m->geom_solref + 2 * g1; const double* r2 = m->geom_solref + 2 * g2;
    if (r1[0] > 0 && r2[0] > 0)
      for (int k = 0; k < 2; k++) c->solref[k] = mix * r1[k] + (1 - mix) * r2[k];
    else
      for (int k = 0; k < 2; k++) c->solref[k] = r1[k] < r2[k] ? r1[k] : r2[k];
    for (int k = 0; k < 5; k++) c->solimp[k] = mix * m->geom_solimp[5 * g1 + k] + (1 - mix) * m->geom_solimp[5 * g2 + k];
  } else {
    const int g = p1 > p2 ? g1 : g2;
    c->dim = m->geom_condim[g];
    memcpy(fri, p1 > p2 ? f1 : f2, sizeof fri);
    memcpy(c->solref, m->geom_solref + 2 * g, sizeof c->solref);
    memcpy(c->solimp, m->geom_solimp + 5 * g, sizeof c->solimp);
  }
  c->friction[0] = c->friction[1] = fri[0]; c->friction[2] = fri[1]; c->friction[3] = c->friction[4] = fri[2];
  for (int k = 0; k < 5; k++) c->friction[k] = fmax(MINMU, c->friction[k]);
  const double margin = fmax(m->geom_margin[g1], m->geom_margin[g2]), gap = fmax(m->geom_gap[g1], m->geom_gap[g2]);
  c->includemargin = margin - gap;
  c->mu = 0;
}

/* closest points of segments p1 + s (q1 - p1) and p2 + t (q2 - p2), s, t in [0, 1] (Ericson, Real-Time Collision
 * Detection 5.1.9); degenerate segments (spheres) included */
static void closest_seg_seg(const double* p1, const double* q1, const double* p2, const double* q2, double* c1, double* c2) {
  double d1[3], d2[3], r[3];
  for (int k = 0; k < 3; k++) { d1[k] = q1[k] - p1[k]; d2[k] = q2[k] - p2[k]; r[k] = p1[k] - p2[k]; }
  const double a = dot3(d1, d1), e = dot3(d2, d2), f = dot3(d2, r), EPS = 1e-12;
  double s, t;
  if (a <= EPS && e <= EPS) { s = t = 0; }
  else if (a <= EPS) { s = 0; t = fmin(fmax(f / e, 0), 1); }
  else {
    const double c = dot3(d1, r);
    if (e <= EPS) { t = 0; s = fmin(fmax(-c / a, 0), 1); }
    else {
      const double b = dot3(d1, d2), den = a * e - b * b;
      s = den > EPS * a * e ? fmin(fmax((b * f - c * e) / den, 0), 1) : 0; /* parallel: any s; 0 */
      t = (b * s + f) / e;
      if (t < 0) { t = 0; s = fmin(fmax(-c / a, 0), 1); }
      else if (t > 1) { t = 1; s = fmin(fmax((b - c) / a, 0), 1); }
    }
  }
  for (int k = 0; k < 3; k++) { c1[k] = p1[k] + s * d1[k]; c2[k] = p2[k] + t * d2[k]; }
}

/* ------------------------------------------------------------------ height field (MuJoCo hfield geom, identity orientation)
 * Grid cell (c, r) is split like MuJoCo's prism strip: vertices (c,r), (c,r+1), (c+1,r), (c+1,r+1), i.e. the diagonal
 * runs from (c,r+1) to (c+1,r).  MuJoCo (mjc_ConvexHField) emits one contact per prism under the geom's bounding box;
 * this restatement keeps ONE contact per robot geom: a sphere its closest triangle (exact point-triangle distance), any
 * other geom its deepest cloud vertex measured against the plane of the triangle under that vertex. */
typedef struct { double a[3], b[3], c[3], n[3]; } HfTri;
static double hf_h(const GqModelDesc* m, int r, int c) { return m->hfield_size[2] * (double)m->hfield_data[r * m->hfield_ncol + c]; }
static void hf_cell_triangle(const GqModelDesc* m, int c, int r, int upper, HfTri* t) {
  const double dx = 2 * m->hfield_size[0] / (m->hfield_ncol - 1), dy = 2 * m->hfield_size[1] / (m->hfield_nrow - 1);
  const double x0 = -m->hfield_size[0] + dx * c, y0 = -m->hfield_size[1] + dy * r, x1 = x0 + dx, y1 = y0 + dy;
  const double h10 = hf_h(m, r, c + 1), h01 = hf_h(m, r + 1, c), hq = upper ? hf_h(m, r + 1, c + 1) : hf_h(m, r, c);
  double gx, gy;
  t->b[0] = x1; t->b[1] = y0; t->b[2] = h10; t->c[0] = x0; t->c[1] = y1; t->c[2] = h01;
  if (!upper) { t->a[0] = x0; t->a[1] = y0; t->a[2] = hq; gx = (h10 - hq) / dx; gy = (h01 - hq) / dy; }
  else { t->a[0] = x1; t->a[1] = y1; t->a[2] = hq; gx = (hq - h01) / dx; gy = (hq - h10) / dy; }
  const double inv = 1 / sqrt(gx * gx + gy * gy + 1);
  t->n[0] = -gx * inv; t->n[1] = -gy * inv; t->n[2] = inv;
}
static int hf_triangle_under(const GqModelDesc* m, double x, double y, HfTri* t) {
  const double dx = 2 * m->hfield_size[0] / (m->hfield_ncol - 1), dy = 2 * m->hfield_size[1] / (m->hfield_nrow - 1);
  const double fx = (x + m->hfield_size[0]) / dx, fy = (y + m->hfield_size[1]) / dy;
  if (!(fx >= 0 && fy >= 0 && fx <= m->hfield_ncol - 1 && fy <= m->hfield_nrow - 1)) return 0;
  int c = (int)fx, r = (int)fy;
  if (c > m->hfield_ncol - 2) c = m->hfield_ncol - 2;
  if (r > m->hfield_nrow - 2) r = m->hfield_nrow - 2;
  hf_cell_triangle(m, c, r, (fx - c) + (fy - r) > 1.0, t);
  return 1;
}
/* closest point of triangle (a, b, c) to p (Ericson, Real-Time Collision Detection 5.1.5) */
static int closest_on_triangle(const double* p, const double* a, const double* b, const double* c, double* q) { /* 1: the projection of p falls inside */
  double ab[3], ac[3], ap[3], bp[3], cp[3];
  for (int k = 0; k < 3; k++) { ab[k] = b[k] - a[k]; ac[k] = c[k] - a[k]; ap[k] = p[k] - a[k]; bp[k] = p[k] - b[k]; cp[k] = p[k] - c[k]; }
  const double d1 = dot3(ab, ap), d2 = dot3(ac, ap);
  if (d1 <= 0 && d2 <= 0) { memcpy(q, a, 24); return 0; }
  const double d3 = dot3(ab, bp), d4 = dot3(ac, bp);
  if (d3 >= 0 && d4 <= d3) { memcpy(q, b, 24); return 0; }
  const double vc = d1 * d4 - d3 * d2;
  if (vc <= 0 && d1 >= 0 && d3 <= 0) { const double v = d1 / (d1 - d3); for (int k = 0; k < 3; k++) q[k] = a[k] + v * ab[k]; return 0; }
  const double d5 = dot3(ab, cp), d6 = dot3(ac, cp);
  if (d6 >= 0 && d5 <= d6) { memcpy(q, c, 24); return 0; }
  const double vb = d5 * d2 - d1 * d6;
  if (vb <= 0 && d2 >= 0 && d6 <= 0) { const double w = d2 / (d2 - d6); for (int k = 0; k < 3; k++) q[k] = a[k] + w * ac[k]; return 0; }
  const double va = d3 * d6 - d5 * d4;
  if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) { const double w = (d4 - d3) / ((d4 - d3) + (d5 - d6)); for (int k = 0; k < 3; k++) q[k] = b[k] + w * (c[k] - b[k]); return 0; }
  const double den = 1 / (va + vb + vc);
  for (int k = 0; k < 3; k++) q[k] = a[k] + vb * den * ab[k] + vc * den * ac[k];
  return 1;
}
/* sphere (centre p in hfield-local coordinates) against the field: signed distance and normal; 0 if farther than reach */
static int sphere_hfield(const GqModelDesc* m, const double* p, double r, double reach, double* dist, double* n) {
  const double dx = 2 * m->hfield_size[0] / (m->hfield_ncol - 1), dy = 2 * m->hfield_size[1] / (m->hfield_nrow - 1), R = r + reach;
  int c0 = (int)floor((p[0] - R + m->hfield_size[0]) / dx), c1 = (int)floor((p[0] + R + m->hfield_size[0]) / dx);
  int r0 = (int)floor((p[1] - R + m->hfield_size[1]) / dy), r1 = (int)floor((p[1] + R + m->hfield_size[1]) / dy);
  if (c0 < 0) c0 = 0;
  if (r0 < 0) r0 = 0;
  if (c1 > m->hfield_ncol - 2) c1 = m->hfield_ncol - 2;
  if (r1 > m->hfield_nrow - 2) r1 = m->hfield_nrow - 2;
  *dist = 1e300; n[0] = n[1] = 0; n[2] = 1;
  for (int rr = r0; rr <= r1; rr++)
    for (int cc = c0; cc <= c1; cc++)
      for (int up = 0; up < 2; up++) {
        HfTri t;
        hf_cell_triangle(m, cc, rr, up, &t);
        double pa[3] = {p[0] - t.a[0], p[1] - t.a[1], p[2] - t.a[2]}, q[3], d[3], dd, nn[3];
        const double side = dot3(pa, t.n);
        if (closest_on_triangle(p, t.a, t.b, t.c, q)) { dd = side - r; memcpy(nn, t.n, sizeof nn); } /* over / under the face */
        else {
          if (side < 0) continue; /* below the plane and outside the column: a neighbour's business */
          for (int k = 0; k < 3; k++) d[k] = p[k] - q[k];
          const double l2 = dot3(d, d);
          if (!(l2 > 1e-12)) continue;
          const double l = sqrt(l2); dd = l - r; for (int k = 0; k < 3; k++) nn[k] = d[k] / l;
        }
        if (dd < *dist) { *dist = dd; memcpy(n, nn, sizeof nn); }
      }
  return *dist < reach;
}

/* ------------------------------------------------------------------ exact narrow phases of primitive pairs (robot-robot and
 * robot - world box): sphere / capsule against box and box against box.  MuJoCo runs mjc_SphereBox, mjc_CapsuleBox (<= 2
 * points) and mjc_BoxBox (<= 8 points) there; those routines are long and their exact point sets cannot be checked here
 * (MuJoCo unavailable), so what is restated is the GEOMETRY they compute - exact closest features, separating-axis
 * penetration - with a contact manifold of at most 2 / 4 points chosen by rules stated below.
 * Conventions as everywhere in MuJoCo: normal from geom 1 to geom 2, point midway between the surfaces, dist = signed
 * distance along the normal (negative: penetration).  The routines return the normal pointing FROM THE BOX (first box) TO THE
 * OTHER geom; callers flip it for the pair's geom order. */
typedef struct { double dist, pos[3], nrm[3], tie; } PairPt; /* tie: margin by which the discrete choices behind the point were made (nearest face of a point inside a box, separating axis): test diagnostics, like Contact.tiegap */

/* signed distance of point p (box frame) to the box of half sizes h, outward normal n (box frame); inside: nearest face */
static double g_point_box_tie; /* set by point_box: gap between the nearest and the second nearest face of a point inside the box (1: outside) */
static double point_box(const double* p, const double* h, double* n) {
  double q[3], d[3], l2 = 0;
  g_point_box_tie = 1.0;
  for (int k = 0; k < 3; k++) { q[k] = fmin(fmax(p[k], -h[k]), h[k]); d[k] = p[k] - q[k]; l2 += d[k] * d[k]; }
  if (l2 > 0) { const double l = sqrt(l2); for (int k = 0; k < 3; k++) n[k] = d[k] / l; return l; }
  int ax = 0; double dmin = 1e300, dsec = 1e300;
  for (int k = 0; k < 3; k++) { const double e = h[k] - fabs(p[k]); if (e < dmin) { dsec = dmin; dmin = e; ax = k; } else if (e < dsec) dsec = e; }
  n[0] = n[1] = n[2] = 0; n[ax] = p[ax] >= 0 ? 1 : -1;
  g_point_box_tie = dsec - dmin;
  return -dmin;
}

/* capsule (axis p0-p1, radius r; a sphere when p0 == p1) against a box (centre bc, axes = columns of bR, half sizes bh).
 * Point 1: the point of the axis closest to the box - the distance of p0 + s (p1 - p0) to the box is convex and piecewise
 * quadratic in s, its derivative piecewise linear with kinks where a coordinate crosses a face plane: evaluated at 0, 1 and
 * the (<= 6) kinks, the root lies between the last kink with a non-positive and the first with a positive derivative (exact).
 * An axis that passes through the box: the middle of the part inside, pushed out through its nearest face.
 * Point 2: the end of the axis farther from point 1, if its own sphere is within the margin of the box (a capsule lying
 * along a face or an edge is carried at both ends, like mjc_CapsuleBox's two-point case). */
static int capsule_box(const double* p0, const double* p1, double r, const double* bc, const double* bR, const double* bh, double margin, PairPt* out) {
  double a[3], b[3], d[3], t0[3], t1[3];
  for (int k = 0; k < 3; k++) { t0[k] = p0[k] - bc[k]; t1[k] = p1[k] - bc[k]; }
  mulmatTvec3(a, bR, t0); mulmatTvec3(b, bR, t1);
  for (int k = 0; k < 3; k++) d[k] = b[k] - a[k];
  double cand[8]; int nc = 0;
  cand[nc++] = 0; cand[nc++] = 1;
  for (int k = 0; k < 3; k++)
    for (int sg = -1; sg <= 1; sg += 2)
      if (fabs(d[k]) > 1e-12) { const double sv = (sg * bh[k] - a[k]) / d[k]; if (sv > 0 && sv < 1) cand[nc++] = sv; }
#define GQO_G(sv, gout, depthout) do { double pp[3], gg = 0, dep = 1e300; for (int k = 0; k < 3; k++) { pp[k] = a[k] + (sv) * d[k]; \
    const double qq = fmin(fmax(pp[k], -bh[k]), bh[k]); gg += d[k] * (pp[k] - qq); dep = fmin(dep, bh[k] - fabs(pp[k])); } gout = gg; depthout = dep; } while (0)
  double g0, g1, dep0, dep1, sstar;
  GQO_G(0.0, g0, dep0); GQO_G(1.0, g1, dep1); (void)dep0; (void)dep1;
  const double epsg = 1e-5 * dot3(d, d); /* an axis (numerically) parallel to the nearest face: the first end, whatever the round-off says */
  if (g0 >= -epsg) sstar = 0;
  else if (g1 <= epsg) sstar = 1;
  else {
    double lo = 0, glo = g0, hi = 1, ghi = g1;
    for (int i = 2; i < nc; i++) {
      double gi, di; GQO_G(cand[i], gi, di); (void)di;
      if (gi < -epsg && cand[i] > lo) { lo = cand[i]; glo = gi; }   /* kinks where the derivative is (numerically) zero bound a flat stretch of */
      if (gi > epsg && cand[i] < hi) { hi = cand[i]; ghi = gi; }     /* equally close points: the secant across it picks one of them, reproducibly */
    }
    sstar = lo + (hi - lo) * (-glo) / (ghi - glo);
  }
  { /* an axis that passes through the box (slab clipping): the middle of the part inside, pushed out through its nearest face */
    double tE = 0, tX = 1; int ok = 1;
    for (int k = 0; k < 3; k++) {
      if (fabs(d[k]) > 1e-12) { const double s1 = (-bh[k] - a[k]) / d[k], s2 = (bh[k] - a[k]) / d[k]; tE = fmax(tE, fmin(s1, s2)); tX = fmin(tX, fmax(s1, s2)); }
      else if (fabs(a[k]) > bh[k]) ok = 0;
    }
    if (ok && tE < tX) sstar = 0.5 * (tE + tX);
  }
#undef GQO_G
  int n = 0;
  for (int q = 0; q < 2; q++) {
    double sv = sstar;
    if (q == 1) { sv = sstar < 0.499 ? 1.0 : 0.0; /* (a geom placed symmetrically has s* = 1/2 up to round-off: not a threshold to sit on) */ if (fabs(sv - sstar) <= 1e-3 || (d[0] == 0 && d[1] == 0 && d[2] == 0)) break; }
    double pp[3], nl[3];
    for (int k = 0; k < 3; k++) pp[k] = a[k] + sv * d[k];
    const double dist = point_box(pp, bh, nl) - r;
    if (dist >= margin) { if (q == 0) return 0; break; }
    PairPt* P = &out[n++];
    double pw[3];
    P->tie = g_point_box_tie;
    mulmatvec3(P->nrm, bR, nl);
    mulmatvec3(pw, bR, pp);
    P->dist = dist;
    for (int k = 0; k < 3; k++) P->pos[k] = bc[k] + pw[k] - P->nrm[k] * (r + 0.5 * dist);
  }
  return n;
}

/* box A against box B (centres, axes = columns of R, half sizes).  Separating-axis test over the 15 axes: the axis of largest
 * separation (least penetration) gives dist and the normal (an edge-edge axis wins only if it beats the best face axis by more
 * than 1e-6 + 5 % of its magnitude - resting boxes sit on faces).  Face axis: the corners of the OTHER box within the margin of
 * the reference face plane whose projection falls inside the reference face (a tolerance of 1e-6) are the contact points, in
 * corner order, the deepest 4 kept; when there are none (the reference face lies inside the other box's face) the corners of
 * the reference face are tested against the other box instead.  Edge axis: one point at the closest points of the two edges.
 * Normal from A to B. */
static int box_box(const double* ca, const double* Ra, const double* ha, const double* cb, const double* Rb, const double* hb, double margin, PairPt* out) {
  double t[3] = {cb[0] - ca[0], cb[1] - ca[1], cb[2] - ca[2]};
  double best = -1e300, second = -1e300, bn[3] = {0, 0, 1}; int bcode = -1;
  double beste = -1e300, en[3] = {0, 0, 1}; int ei = -1, ej = -1;
  const double* R[2] = {Ra, Rb};
  for (int w = 0; w < 2; w++)
    for (int i = 0; i < 3; i++) {
      double L[3] = {R[w][i], R[w][3 + i], R[w][6 + i]};
      double tl = dot3(t, L), ra = 0, rb = 0;
      for (int j = 0; j < 3; j++) {
        ra += ha[j] * fabs(Ra[j] * L[0] + Ra[3 + j] * L[1] + Ra[6 + j] * L[2]);
        rb += hb[j] * fabs(Rb[j] * L[0] + Rb[3 + j] * L[1] + Rb[6 + j] * L[2]);
      }
      const double sep = fabs(tl) - ra - rb;
      if (sep > best + 2e-6) { second = best; best = sep; bcode = 3 * w + i; for (int k = 0; k < 3; k++) bn[k] = tl >= 0 ? L[k] : -L[k]; } /* a later axis must win by more than fp32 noise */
      else if (sep > second) second = sep;
    }
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double A[3] = {Ra[i], Ra[3 + i], Ra[6 + i]}, B[3] = {Rb[j], Rb[3 + j], Rb[6 + j]}, L[3];
      cross3(L, A, B);
      const double ln = sqrt(dot3(L, L));
      if (ln < 0.1) continue; /* (nearly) parallel edges: the closest points along them are ill-conditioned and the face axes describe the contact */
      for (int k = 0; k < 3; k++) L[k] /= ln;
      double tl = dot3(t, L), ra = 0, rb = 0;
      for (int q = 0; q < 3; q++) {
        ra += ha[q] * fabs(Ra[q] * L[0] + Ra[3 + q] * L[1] + Ra[6 + q] * L[2]);
        rb += hb[q] * fabs(Rb[q] * L[0] + Rb[3 + q] * L[1] + Rb[6 + q] * L[2]);
      }
      const double sep = fabs(tl) - ra - rb;
      if (sep > beste + 2e-6) { beste = sep; ei = i; ej = j; for (int k = 0; k < 3; k++) en[k] = tl >= 0 ? L[k] : -L[k]; }
    }
  const int edge = ei >= 0 && beste > best + 1e-6 + 0.05 * fabs(best);
  const double sep = edge ? beste : best;
  if (sep >= margin) return 0;
  if (edge) { /* closest points of edge i of A and edge j of B: the edges nearest the other box along the axis */
    double pa[3], pb[3], ua[3] = {Ra[ei], Ra[3 + ei], Ra[6 + ei]}, ub[3] = {Rb[ej], Rb[3 + ej], Rb[6 + ej]};
    for (int k = 0; k < 3; k++) { pa[k] = ca[k]; pb[k] = cb[k]; }
    for (int q = 0; q < 3; q++) {
      if (q != ei) { const double dq = Ra[q] * en[0] + Ra[3 + q] * en[1] + Ra[6 + q] * en[2], sg = fabs(dq) < 1e-4 ? 0 : (dq >= 0 ? 1 : -1); for (int k = 0; k < 3; k++) pa[k] += sg * ha[q] * Ra[3 * k + q]; }
      if (q != ej) { const double dq = Rb[q] * en[0] + Rb[3 + q] * en[1] + Rb[6 + q] * en[2], sg = fabs(dq) < 1e-4 ? 0 : (dq >= 0 ? -1 : 1); for (int k = 0; k < 3; k++) pb[k] += sg * hb[q] * Rb[3 * k + q]; }
    }
    /* lines pa + sa ua, pb + sb ub */
    double dp[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
    const double uaub = dot3(ua, ub), q1 = dot3(ua, dp), q2 = -dot3(ub, dp), den = 1 - uaub * uaub;
    double sa = den > 1e-12 ? (q1 + uaub * q2) / den : 0, sb = den > 1e-12 ? (uaub * q1 + q2) / den : 0;
    sa = fmin(fmax(sa, -ha[ei]), ha[ei]); sb = fmin(fmax(sb, -hb[ej]), hb[ej]);
    PairPt* P = &out[0];
    P->dist = sep; P->tie = fabs(beste - (best + 1e-6 + 0.05 * fabs(best)));
    for (int k = 0; k < 3; k++) { P->nrm[k] = en[k]; P->pos[k] = 0.5 * ((pa[k] + sa * ua[k]) + (pb[k] + sb * ub[k])); }
    return 1;
  }
  /* face axis: reference box (the one that owns the axis) and the other ("incident") box */
  const int refB = bcode >= 3, ax = bcode % 3;
  const double* cr = refB ? cb : ca; const double* Rr = refB ? Rb : Ra; const double* hr = refB ? hb : ha;
  const double* ci = refB ? ca : cb; const double* Ri = refB ? Ra : Rb; const double* hi = refB ? ha : hb;
  double nr[3]; /* outward normal of the reference face, pointing at the incident box */
  for (int k = 0; k < 3; k++) nr[k] = refB ? -bn[k] : bn[k];
  double cd[8], cp[8][3]; int ncand = 0;
  for (int pass = 0; pass < 2 && ncand == 0; pass++) {
    /* pass 0: corners of the incident box against the reference face; pass 1: corners of the reference face against the incident box */
    for (int v = 0; v < 8; v++) {
      double loc[3] = {(v & 1) ? 1.0 : -1.0, (v & 2) ? 1.0 : -1.0, (v & 4) ? 1.0 : -1.0}, w[3];
      if (pass == 0) {
        for (int k = 0; k < 3; k++) w[k] = ci[k] + Ri[3 * k] * loc[0] * hi[0] + Ri[3 * k + 1] * loc[1] * hi[1] + Ri[3 * k + 2] * loc[2] * hi[2];
        double rel[3] = {w[0] - cr[0], w[1] - cr[1], w[2] - cr[2]}, lr[3];
        mulmatTvec3(lr, Rr, rel);
        const double sgn = (Rr[ax] * nr[0] + Rr[3 + ax] * nr[1] + Rr[6 + ax] * nr[2]) >= 0 ? 1 : -1;
        const double dd = sgn * lr[ax] - hr[ax];
        int inside = 1;
        for (int k = 0; k < 3; k++) if (k != ax && fabs(lr[k]) > hr[k] + 1e-6) inside = 0;
        if (dd < margin && inside) { cd[ncand] = dd; for (int k = 0; k < 3; k++) cp[ncand][k] = w[k] - nr[k] * 0.5 * dd; ncand++; }
      } else {
        const double sgn = (Rr[ax] * nr[0] + Rr[3 + ax] * nr[1] + Rr[6 + ax] * nr[2]) >= 0 ? 1 : -1;
        if (loc[ax] * sgn < 0) continue; /* only the four corners of the reference face */
        for (int k = 0; k < 3; k++) w[k] = cr[k] + Rr[3 * k] * loc[0] * hr[0] + Rr[3 * k + 1] * loc[1] * hr[1] + Rr[3 * k + 2] * loc[2] * hr[2];
        double rel[3] = {w[0] - ci[0], w[1] - ci[1], w[2] - ci[2]}, li[3], nl[3];
        mulmatTvec3(li, Ri, rel);
        const double dd = point_box(li, hi, nl);
        if (dd < margin) { cd[ncand] = dd; for (int k = 0; k < 3; k++) cp[ncand][k] = w[k] + nr[k] * 0.5 * dd; ncand++; }
      }
    }
  }
  double mincd = 1e300;
  for (int v = 0; v < ncand; v++) mincd = fmin(mincd, cd[v]);
  if (ncand == 0 || mincd > sep + 1e-4) { /* no corner carries the penetration the axis test found (faces crossing, edges poking through):
                                           * one more point at the support of the incident box, at the axis depth */
    double w[3] = {ci[0], ci[1], ci[2]};
    for (int q = 0; q < 3; q++) { /* support point; an axis (numerically) parallel to the face contributes its midpoint */
      const double dq = Ri[q] * nr[0] + Ri[3 + q] * nr[1] + Ri[6 + q] * nr[2], sg = fabs(dq) < 1e-4 ? 0 : (dq >= 0 ? -1 : 1);
      for (int k = 0; k < 3; k++) w[k] += sg * hi[q] * Ri[3 * k + q];
    }
    if (ncand == 8) ncand = 7;
    cd[ncand] = sep; for (int k = 0; k < 3; k++) cp[ncand][k] = w[k] - nr[k] * 0.5 * sep;
    ncand++;
  }
  /* the deepest 4, in candidate order */
  int keep[8], nk = 0;
  for (int v = 0; v < ncand; v++) keep[v] = 1;
  for (int drop = ncand; drop > 4; drop--) { int worst = -1; double wv = -1e300; for (int v = 0; v < ncand; v++) if (keep[v] && cd[v] >= wv) { wv = cd[v]; worst = v; } keep[worst] = 0; }
  for (int v = 0; v < ncand; v++)
    if (keep[v]) { PairPt* P = &out[nk++]; P->dist = cd[v]; P->tie = fmin(fabs(best - second - 2e-6), ei >= 0 ? fabs(beste - (best + 1e-6 + 0.05 * fabs(best))) : 1.0); for (int k = 0; k < 3; k++) { P->pos[k] = cp[v][k]; P->nrm[k] = bn[k]; } }
  return nk;
}

/* test hooks (tests/test_oracle_invariants.py checks the pair routines against brute-force geometry): out = n x [dist, pos[3], nrm[3]] */
int gqo_test_capsule_box(const double* p0, const double* p1, double r, const double* bc, const double* bR, const double* bh, double margin, double* out) {
  PairPt pts[4];
  const int n = capsule_box(p0, p1, r, bc, bR, bh, margin, pts);
  for (int q = 0; q < n; q++) { out[7 * q] = pts[q].dist; memcpy(out + 7 * q + 1, pts[q].pos, 24); memcpy(out + 7 * q + 4, pts[q].nrm, 24); }
  return n;
}
int gqo_test_box_box(const double* ca, const double* Ra, const double* ha, const double* cb, const double* Rb, const double* hb, double margin, double* out) {
  PairPt pts[4];
  const int n = box_box(ca, Ra, ha, cb, Rb, hb, margin, pts);
  for (int q = 0; q < n; q++) { out[7 * q] = pts[q].dist; memcpy(out + 7 * q + 1, pts[q].pos, 24); memcpy(out + 7 * q + 4, pts[q].nrm, 24); }
  return n;
}

/* counters of the convex routine's work (tools/convex_census.py): calls, contacts, GJK iterations, EPA runs, EPA iterations; not thread-safe */
static long long g_cvx_stat[12];
static long long g_cvx_hist[2][64]; /* census: calls by GJK iterations / contacts by EPA iterations */
void gqo_cvx_hist(long long* out, int reset) { if (out) memcpy(out, g_cvx_hist, sizeof g_cvx_hist); if (reset) memset(g_cvx_hist, 0, sizeof g_cvx_hist); }
void gqo_cvx_stats(long long* out, int reset) { if (out) memcpy(out, g_cvx_stat, sizeof g_cvx_stat); if (reset) memset(g_cvx_stat, 0, sizeof g_cvx_stat); }
/* test diagnostics (Contact.tiegap): is the contact POINT of a convex pair determined?  Depth and normal of the minimum translation are
 * unique, but where two faces, a face and an edge or two parallel edges meet, every point of their overlap is a valid witness and the
 * polytope's last triangle - i.e. the iteration path, which round-off steers - picks one.  Dimension of the support sets along +-n
 * (vertices within 1e-7 of the support planes): ambiguous when they add up to three or more, or are two parallel edges. */
static int cvx_support_set(const Cvx* s, const double* d, double axis[3]) {
  double best = -1e300, W[8][3], first[3] = {0, 0, 0}, far2 = 0;
  const int nv = s->box ? 8 : s->nv;
  int cnt = 0;
  axis[0] = axis[1] = axis[2] = 0;
  for (int pass = 0; pass < 2; pass++)
    for (int v = 0; v < nv; v++) {
      double q[3], w[3];
      if (s->box) { for (int k = 0; k < 3; k++) q[k] = ((v >> k) & 1) ? -s->h[k] : s->h[k]; } else memcpy(q, s->V + 3 * v, sizeof q);
      for (int k = 0; k < 3; k++) w[k] = s->t[k] + s->R[3 * k] * q[0] + s->R[3 * k + 1] * q[1] + s->R[3 * k + 2] * q[2];
      const double pr = cvx_dot(w, d);
      if (pass == 0) { if (pr > best) best = pr; continue; }
      if (pr < best - 1e-7) continue;
      if (cnt == 0) memcpy(first, w, sizeof first);
      else { const double e[3] = {w[0] - first[0], w[1] - first[1], w[2] - first[2]}; const double l2 = cvx_dot(e, e); if (l2 > far2) { far2 = l2; memcpy(axis, e, sizeof e); } }
      if (cnt < 8) memcpy(W[cnt], w, sizeof w);
      cnt++;
    }
  if (cnt <= 1 || far2 < 1e-16) return 0;
  /* more than an edge: some support vertex off the line through the first one along `axis` */
  for (int i = 1; i < (cnt < 8 ? cnt : 8); i++) {
    const double e[3] = {W[i][0] - first[0], W[i][1] - first[1], W[i][2] - first[2]};
    double c[3];
    cvx_cross(c, e, axis);
    if (cvx_dot(c, c) > 1e-14 * far2) return 2;
  }
  return cnt > 8 ? 2 : 1;
}
static double cvx_point_tie(const Cvx* A, const Cvx* B, const double* n) {
  double ea[3], eb[3], nn[3] = {-n[0], -n[1], -n[2]};
  const int da = cvx_support_set(A, n, ea), db = cvx_support_set(B, nn, eb);
  if (da + db >= 3) return 0.0;
  if (da == 1 && db == 1) { double c[3]; cvx_cross(c, ea, eb); if (cvx_dot(c, c) < 1e-8 * cvx_dot(ea, ea) * cvx_dot(eb, eb)) return 0.0; }
  return 1.0;
}
static int g_cvx_last_queries;
static int g_cvx_capped; /* the last pair ran into an iteration cap: its answer is the iteration's state there, not the converged one (test diagnostics) */
static int cvx_pair_counted(const Cvx* A, const Cvx* B, double margin, double* dist, double* nrm, double* pos, int self, const double* hint) {
  int it[2] = {0, 0};
  const int rc = cvx_pair(A, B, margin, dist, nrm, pos, it, hint);
  long long* s = g_cvx_stat + (self ? 4 : 0);
  s[0] += 1; s[1] += rc; s[2] += it[0]; s[3] += it[1];
  g_cvx_last_queries = 1 + it[0] + it[1];
  g_cvx_hist[0][it[0] < 63 ? it[0] : 63]++; if (rc && it[1] > 0) g_cvx_hist[1][it[1] < 63 ? it[1] : 63]++;
  g_cvx_capped = rc && (it[1] >= CVX_EPA_MAXIT || it[0] >= CVX_GJK_MAXIT);
  g_cvx_stat[9] += g_cvx_capped;
  return rc;
}

/* mid phase of a convex pair: the clouds' geom-frame boxes, taken to the world, are held apart by more than `reach` along one of the
 * 15 separating-axis candidates -> the hulls inside them are too (conservative: never hides a contact).  What the kernel's lane = pair
 * pass does in front of its wave-serial GJK (csrc/gq_convex.h obb_apart). */
static const double* cloud_box(GqOracle* o, int cl) {
  const GqModelDesc* m = &o->d;
  if (cl < 64 && !o->cloud_aabb_ok[cl]) {
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int v = 0; v < m->cloud_vertnum[cl]; v++)
      for (int k = 0; k < 3; k++) { const double c = m->vert_pos[3 * (m->cloud_vertadr[cl] + v) + k]; if (c < lo[k]) lo[k] = c; if (c > hi[k]) hi[k] = c; }
    for (int k = 0; k < 3; k++) { o->cloud_aabb[cl][k] = 0.5 * (lo[k] + hi[k]); o->cloud_aabb[cl][3 + k] = 0.5 * (hi[k] - lo[k]) * 1.0001 + 1e-7; }
    o->cloud_aabb_ok[cl] = 1;
  }
  return o->cloud_aabb[cl < 64 ? cl : 63];
}
static int obb_apart(const double* ca, const double* Ra, const double* ha, const double* cb, const double* Rb, const double* hb, double reach, double* axis) {
  double C[3][3], AC[3][3], t[3], d[3] = {cb[0] - ca[0], cb[1] - ca[1], cb[2] - ca[2]};
  for (int i = 0; i < 3; i++) {
    t[i] = Ra[i] * d[0] + Ra[3 + i] * d[1] + Ra[6 + i] * d[2];
    for (int j = 0; j < 3; j++) { C[i][j] = Ra[i] * Rb[j] + Ra[3 + i] * Rb[3 + j] + Ra[6 + i] * Rb[6 + j]; AC[i][j] = fabs(C[i][j]) + 1e-9; }
  }
  /* the candidate with the largest gap (first of equals, in this order) is GJK's first search direction, turned to point from A to B */
  double best = -1e300;
  int apart = 0;
  axis[0] = d[0]; axis[1] = d[1]; axis[2] = d[2];
  for (int i = 0; i < 3; i++) {
    const double gap = fabs(t[i]) - (ha[i] + hb[0] * AC[i][0] + hb[1] * AC[i][1] + hb[2] * AC[i][2]);
    if (gap > reach) apart = 1;
    if (gap > best) { best = gap; const double sg = t[i] < 0 ? -1 : 1; for (int k = 0; k < 3; k++) axis[k] = sg * Ra[3 * k + i]; }
  }
  for (int j = 0; j < 3; j++) {
    const double tj = t[0] * C[0][j] + t[1] * C[1][j] + t[2] * C[2][j];
    const double gap = fabs(tj) - (hb[j] + ha[0] * AC[0][j] + ha[1] * AC[1][j] + ha[2] * AC[2][j]);
    if (gap > reach) apart = 1;
    if (gap > best) { best = gap; const double sg = tj < 0 ? -1 : 1; for (int k = 0; k < 3; k++) axis[k] = sg * Rb[3 * k + j]; }
  }
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      const double len2 = 1.0 - C[i][j] * C[i][j]; /* |a_i x b_j|^2 */
      if (len2 < 1e-4) continue;                    /* nearly parallel edges: the face axes decide */
      const double len = sqrt(len2), tp = t[i2] * C[i1][j] - t[i1] * C[i2][j];
      const double gap = (fabs(tp) - (ha[i1] * AC[i2][j] + ha[i2] * AC[i1][j] + hb[j1] * AC[i][j2] + hb[j2] * AC[i][j1])) / len;
      if (gap > reach) apart = 1;
      if (gap > best) {
        best = gap;
        const double ai[3] = {Ra[i], Ra[3 + i], Ra[6 + i]}, bj[3] = {Rb[j], Rb[3 + j], Rb[6 + j]};
        double c[3];
        cross3(c, ai, bj);
        const double sg = dot3(c, d) < 0 ? -1 : 1;
        for (int k = 0; k < 3; k++) axis[k] = sg * c[k] / len;
      }
    }
  return apart;
}

/* test hook (tests/test_oracle_invariants.py): the convex routine on two shapes - clouds (h = NULL) or analytic boxes (V = NULL).
 * out: dist, pos[3], nrm[3], GJK iterations, EPA iterations */
int gqo_test_convex(const double* VA, int na, const double* hA, const double* RA, const double* tA, double rA,
                    const double* VB, int nb, const double* hB, const double* RB, const double* tB, double rB, double margin, double* out) {
  Cvx A, B;
  memset(&A, 0, sizeof A); memset(&B, 0, sizeof B);
  A.box = VA == NULL; A.V = VA; A.nv = na; A.r = rA; memcpy(A.R, RA, sizeof A.R); memcpy(A.t, tA, sizeof A.t); if (hA) memcpy(A.h, hA, sizeof A.h);
  B.box = VB == NULL; B.V = VB; B.nv = nb; B.r = rB; memcpy(B.R, RB, sizeof B.R); memcpy(B.t, tB, sizeof B.t); if (hB) memcpy(B.h, hB, sizeof B.h);
  int it[2] = {0, 0};
  const int rc = cvx_pair(&A, &B, margin, out, out + 4, out + 1, it, NULL);
  out[7] = it[0]; out[8] = it[1];
  return rc;
}

/* a robot collision geom as a primitive for the pair routines: 1 sphere / capsule (world end points, radius), 2 box (world
 * centre, axes, half sizes), 0 anything else (cylinders, hulls: capsule proxy / vertex cloud) */
typedef struct { int kind; double p0[3], p1[3], r, c[3], R[9], h[3]; } Prim;
static void prim_of_geom(const GqOracle* o, int g, Prim* P) {
  const GqModelDesc* m = &o->d;
  const int cl = m->geom_cloudid[g], nv = m->cloud_vertnum[cl], type = m->geom_type ? m->geom_type[g] : (nv == 1 ? 2 : (nv == 2 ? 3 : 7));
  const double* V = m->vert_pos + 3 * m->cloud_vertadr[cl];
  P->kind = 0;
  if ((type == 2 && nv == 1) || (type == 3 && nv == 2)) {
    P->kind = 1; P->r = m->cloud_radius[cl];
    mulmatvec3(P->p0, o->geom_xmat[g], V); mulmatvec3(P->p1, o->geom_xmat[g], V + 3 * (nv - 1));
    for (int k = 0; k < 3; k++) { P->p0[k] += o->geom_xpos[g][k]; P->p1[k] += o->geom_xpos[g][k]; }
  } else if (type == 6 && nv == 8) {
    P->kind = 2;
    memcpy(P->c, o->geom_xpos[g], sizeof P->c); memcpy(P->R, o->geom_xmat[g], sizeof P->R);
    for (int k = 0; k < 3; k++) P->h[k] = V[21 + k];
  }
}

static void gqo_collision(GqOracle* o) {
  const GqModelDesc* m = &o->d;
  o->ncon = 0;
  const double normal[3] = {0, 0, 1};
  for (int g = 0; g < m->ngeom && o->ncon < NCON; g++) {
    int cl = m->geom_cloudid[g];
    if (cl < 0 || m->geom_bodyid[g] == 0) continue;
    double margin = m->floor_margin > m->geom_margin[g] ? m->floor_margin : m->geom_margin[g];
    double r = m->cloud_radius[cl];
    /* bounding-sphere cull (mj broadphase equivalent for a plane) */
    if (o->geom_xpos[g][2] - m->geom_rbound[g] > margin) continue;
    const int nv = m->cloud_vertnum[cl], type = m->geom_type ? m->geom_type[g] : (nv == 1 ? 2 : (nv == 2 ? 3 : 7));
    const double* gp = o->geom_xpos[g]; const double* gm = o->geom_xmat[g];
    const double* V = m->vert_pos + 3 * m->cloud_vertadr[cl];
    double pdist[4], ppos[4][3], hint[3] = {0, 0, 0}, tiegap = 1.0;
    int np = 0, use_hint = 0;
    if (type == 3 && nv == 2) { /* mjraw_PlaneCapsule: the cloud holds (-axis end, +axis end); MuJoCo tests the + end first */
      for (int q = 0; q < 2; q++) {
        const int v = 1 - q;
        double w[3];
        mulmatvec3(w, gm, V + 3 * v);
        const double dist = w[2] + gp[2] - r;
        if (dist >= margin) continue;
        pdist[np] = dist;
        for (int k = 0; k < 3; k++) ppos[np][k] = w[k] + gp[k] - normal[k] * (r + 0.5 * dist);
        np++;
      }
      hint[0] = gm[2]; hint[1] = gm[5]; hint[2] = gm[8]; use_hint = 1;
    } else if (type == 6 && nv == 8) { /* mjraw_PlaneBox */
      const double dist0 = gp[2];
      for (int v = 0; v < 8 && np < 4; v++) {
        double w[3];
        mulmatvec3(w, gm, V + 3 * v);
        const double ldist = w[2];
        if (dist0 + ldist >= margin || ldist > 0) continue;
        pdist[np] = dist0 + ldist;
        for (int k = 0; k < 3; k++) ppos[np][k] = w[k] + gp[k] - normal[k] * 0.5 * pdist[np];
        np++;
      }
    } else if (type == 5 && nv == 32) { /* mjc_PlaneCylinder; radius and half length from the prism the cloud holds */
      const double rad = sqrt(V[0] * V[0] + V[1] * V[1]), hl = fabs(V[2]);
      double axis[3] = {gm[2], gm[5], gm[8]}, vec[3], prjaxis = axis[2];
      if (prjaxis > 0) { for (int k = 0; k < 3; k++) axis[k] = -axis[k]; prjaxis = -prjaxis; }
      const double dist0 = gp[2];
      for (int k = 0; k < 3; k++) vec[k] = axis[k] * prjaxis - normal[k];
      const double len2 = dot3(vec, vec);
      if (len2 >= MINVAL * MINVAL) { const double sc = rad / sqrt(len2); for (int k = 0; k < 3; k++) vec[k] *= sc; }
      else { vec[0] = gm[0] * rad; vec[1] = gm[3] * rad; vec[2] = gm[6] * rad; } /* disk parallel to the plane: the cylinder's x axis */
      const double prjvec = vec[2];
      for (int k = 0; k < 3; k++) axis[k] *= hl;
      prjaxis *= hl;
      if (dist0 + prjaxis + prjvec < margin) {
        pdist[np] = dist0 + prjaxis + prjvec;
        for (int k = 0; k < 3; k++) ppos[np][k] = gp[k] + vec[k] + axis[k] - normal[k] * 0.5 * pdist[np];
        np++;
        if (dist0 - prjaxis + prjvec < margin) {
          pdist[np] = dist0 - prjaxis + prjvec;
          for (int k = 0; k < 3; k++) ppos[np][k] = gp[k] + vec[k] - axis[k] - normal[k] * 0.5 * pdist[np];
          np++;
        }
        const double prjvec1 = -0.5 * prjvec;
        if (dist0 + prjaxis + prjvec1 < margin) { /* two more points of the near cap: the triangle's other corners */
          double vec1[3];
          cross3(vec1, vec, axis);
          const double n1 = sqrt(dot3(vec1, vec1));
          if (n1 < MINVAL) { vec1[0] = 1; vec1[1] = 0; vec1[2] = 0; } else for (int k = 0; k < 3; k++) vec1[k] /= n1;
          for (int k = 0; k < 3; k++) vec1[k] *= rad * sqrt(3.0) / 2;
          for (int sgn = 1; sgn >= -1; sgn -= 2) {
            pdist[np] = dist0 + prjaxis + prjvec1;
            for (int k = 0; k < 3; k++) ppos[np][k] = gp[k] + sgn * vec1[k] + axis[k] - 0.5 * vec[k] - normal[k] * 0.5 * pdist[np];
            np++;
          }
        }
      }
    } else { /* sphere: exact; mesh (and anything else): the support vertex of the cloud */
      double best = 1e300, second = 1e300, bw[3] = {0, 0, 0};
      int vbest = 0;
      for (int v = 0; v < nv; v++) {
        double w[3];
        mulmatvec3(w, gm, V + 3 * v);
        const double dv = w[2] + gp[2] - r;
        if (dv < best) { second = best; best = dv; vbest = v; for (int k = 0; k < 3; k++) bw[k] = w[k] + gp[k]; }
        else if (dv < second) second = dv;
      }
      if (best < margin) {
        pdist[0] = best;
        for (int k = 0; k < 3; k++) ppos[0][k] = bw[k] - normal[k] * (r + 0.5 * best);
        np = 1;
        tiegap = second - best;
        /* mjc_PlaneConvex, mesh geoms: after the support vertex, the vertices the hull graph joins to it are tried in the graph's
         * order and those within the margin become contacts too, until the pair has three (GqModelDesc.vert_adj*: the hull's edge
         * graph; neighbours by ascending vertex index) */
        if (m->vert_adjnum && m->vert_adjadr && m->vert_adj) {
          const int va = m->cloud_vertadr[cl] + vbest;
          for (int q = 0; q < m->vert_adjnum[va] && np < 3; q++) {
            const int u = m->vert_adj[m->vert_adjadr[va] + q];
            double w[3];
            mulmatvec3(w, gm, V + 3 * u);
            const double du = w[2] + gp[2] - r;
            if (fabs(du - margin) < tiegap) tiegap = fabs(du - margin);
            if (du >= margin) continue;
            pdist[np] = du;
            for (int k = 0; k < 3; k++) ppos[np][k] = w[k] + gp[k] - normal[k] * (r + 0.5 * du);
            np++;
          }
          /* a tie for the support vertex changes the whole manifold, a neighbour at the margin changes its size: both are 'tie' envs */
        }
      }
    }
    for (int q = 0; q < np && o->ncon < NCON; q++) {
      Contact* c = &o->contact[o->ncon++];
      c->geom = g; c->body = m->geom_bodyid[g]; c->geom1 = -1; c->body1 = 0;
      c->dist = pdist[q];
      c->tiegap = tiegap;
      memcpy(c->pos, ppos[q], sizeof c->pos);
      set_frame(c, normal, use_hint ? hint : NULL);
      contact_param(o, -1, g, c);
    }
  }
  /* world boxes (geoms 1..nbox of the world body).  Sphere geoms: exact sphere-box distance (mjc_SphereBox: clamp the
   * centre into the box; inside, leave through the nearest face).  Every other robot geom: the same test for each vertex
   * of its cloud inflated by the cloud radius, deepest one kept - one contact per (box, geom) pair.  NOT MuJoCo's
   * mesh-box routine (libccd penetration on the convex hulls): an approximation that is exact for vertex-face touching. */
  for (int w = 0; w < m->nbox && o->ncon < NCON; w++) { /* (box, geom id) order, as the kernel appends them */
    for (int g = 0; g < m->ngeom && o->ncon < NCON; g++) {
      int cl = m->geom_cloudid[g];
      if (cl < 0 || m->geom_bodyid[g] == 0) continue;
      const double r = m->cloud_radius[cl];
      const double* bp = m->box_pos + 3 * w; const double* bm = m->box_mat + 9 * w; const double* bs = m->box_size + 3 * w;
      const double margin = m->box_margin[w] > m->geom_margin[g] ? m->box_margin[w] : m->geom_margin[g];
      double dc[3] = {o->geom_xpos[g][0] - bp[0], o->geom_xpos[g][1] - bp[1], o->geom_xpos[g][2] - bp[2]};
      const double brad = sqrt(bs[0] * bs[0] + bs[1] * bs[1] + bs[2] * bs[2]);
      if (sqrt(dot3(dc, dc)) > brad + m->geom_rbound[g] + margin) continue; /* bounding spheres */
      Prim P;
      prim_of_geom(o, g, &P);
      if (P.kind != 0 && !is_foot(m, g)) { /* capsule / box geoms of the robot: exact pair routines (feet keep the sphere path below) */
        PairPt pts[4];
        const int np = P.kind == 1 ? capsule_box(P.p0, P.p1, P.r, bp, bm, bs, margin, pts) : box_box(bp, bm, bs, P.c, P.R, P.h, margin, pts);
        for (int q = 0; q < np && o->ncon < NCON; q++) {
          Contact* c = &o->contact[o->ncon++];
          c->geom = g; c->body = m->geom_bodyid[g]; c->geom1 = -1; c->body1 = 0; c->dist = pts[q].dist; c->tiegap = pts[q].tie;
          memcpy(c->pos, pts[q].pos, sizeof c->pos);
          set_frame(c, pts[q].nrm, NULL);
          contact_param(o, w, g, c);
        }
        continue;
      }
      if (!is_foot(m, g)) { /* hull / cylinder clouds: MuJoCo's general convex routine (mjc_Convex), one contact per pair - gq_convex.h */
        Cvx A, B;
        memset(&A, 0, sizeof A); memset(&B, 0, sizeof B);
        A.box = 1; memcpy(A.R, bm, sizeof A.R); memcpy(A.t, bp, sizeof A.t); memcpy(A.h, bs, sizeof A.h);
        B.V = m->vert_pos + 3 * m->cloud_vertadr[cl]; B.nv = m->cloud_vertnum[cl]; B.r = r;
        memcpy(B.R, o->geom_xmat[g], sizeof B.R); memcpy(B.t, o->geom_xpos[g], sizeof B.t);
        double dist, nrm[3], pos[3];
        double hint[3];
        { /* GJK's first direction: out of the box towards the centre of the cloud's box (csrc/gq_boxes.h box_item_scan) */
          const double* cb = cloud_box(o, cl);
          double cw[3], cl3[3], nl[3];
          mulmatvec3(cw, o->geom_xmat[g], cb);
          for (int k = 0; k < 3; k++) cw[k] += o->geom_xpos[g][k] - bp[k];
          mulmatTvec3(cl3, bm, cw);
          point_box(cl3, bs, nl);
          mulmatvec3(hint, bm, nl);
        }
        if (!cvx_pair_counted(&A, &B, margin, &dist, nrm, pos, 0, hint)) continue;
        Contact* c = &o->contact[o->ncon++];
        c->geom = g; c->body = m->geom_bodyid[g]; c->geom1 = -1; c->body1 = 0; c->dist = dist; c->tiegap = g_cvx_capped ? 0.0 : (cvx_point_tie(&A, &B, nrm) > 0 ? 1.0 : -1.0); /* -1: depth and normal are determined, the contact POINT is not */
        memcpy(c->pos, pos, sizeof c->pos);
        set_frame(c, nrm, NULL);
        contact_param(o, w, g, c);
        continue;
      }
      double best = 1e300, second = 1e300, bn[3] = {0, 0, 1}, bv[3] = {0, 0, 0};
      for (int v = 0; v < m->cloud_vertnum[cl]; v++) {
        double wv[3], lc[3], q[3], nl[3], dist;
        mulmatvec3(wv, o->geom_xmat[g], m->vert_pos + 3 * (m->cloud_vertadr[cl] + v));
        for (int k = 0; k < 3; k++) wv[k] += o->geom_xpos[g][k] - bp[k];
        mulmatTvec3(lc, bm, wv);                       /* vertex in the box frame */
        int inside = 1;
        for (int k = 0; k < 3; k++) { q[k] = fmin(fmax(lc[k], -bs[k]), bs[k]); if (q[k] != lc[k]) inside = 0; }
        if (inside) { /* leave through the nearest face */
          int ax = 0; double dmin = 1e300;
          for (int k = 0; k < 3; k++) { double dk = bs[k] - fabs(lc[k]); if (dk < dmin) { dmin = dk; ax = k; } }
          nl[0] = nl[1] = nl[2] = 0; nl[ax] = lc[ax] >= 0 ? 1 : -1;
          dist = -dmin - r;
        } else {
          double dd[3] = {lc[0] - q[0], lc[1] - q[1], lc[2] - q[2]}, len = sqrt(dot3(dd, dd));
          for (int k = 0; k < 3; k++) nl[k] = dd[k] / len;
          dist = len - r;
        }
        if (dist < best) {
          second = best; best = dist;
          mulmatvec3(bn, bm, nl);
          for (int k = 0; k < 3; k++) bv[k] = wv[k] + bp[k];
        } else if (dist < second) second = dist;
      }
      if (best >= margin) continue;
      Contact* c = &o->contact[o->ncon++];
      c->geom = g; c->body = m->geom_bodyid[g]; c->geom1 = -1; c->body1 = 0; c->dist = best; c->tiegap = second - best;
      for (int k = 0; k < 3; k++) c->pos[k] = bv[k] - bn[k] * (r + 0.5 * best); /* midway between the surfaces */
      set_frame(c, bn, NULL);
      contact_param(o, w, g, c);
    }
  }
  /* the height field: world geom after the boxes */
  if (m->hfield_nrow > 0)
    for (int g = 0; g < m->ngeom && o->ncon < NCON; g++) {
      int cl = m->geom_cloudid[g];
      if (cl < 0 || m->geom_bodyid[g] == 0) continue;
      const double r = m->cloud_radius[cl];
      const double margin = m->hfield_margin > m->geom_margin[g] ? m->hfield_margin : m->geom_margin[g];
      double best = 1e300, second = 1e300, bn[3] = {0, 0, 1}, bv[3] = {0, 0, 0};
      if (is_foot(m, g)) { /* foot sphere (the kernel's other collision items are vertex clouds, spheres included) */
        double wv[3], lc[3], d, n[3];
        mulmatvec3(wv, o->geom_xmat[g], m->vert_pos + 3 * m->cloud_vertadr[cl]);
        for (int k = 0; k < 3; k++) { wv[k] += o->geom_xpos[g][k]; lc[k] = wv[k] - m->hfield_pos[k]; }
        if (!sphere_hfield(m, lc, r, fmax(margin, 0) + 1e-4, &d, n)) continue;
        best = d; memcpy(bn, n, sizeof bn); memcpy(bv, wv, sizeof bv);
      } else
        for (int v = 0; v < m->cloud_vertnum[cl]; v++) {
          double wv[3], lc[3];
          mulmatvec3(wv, o->geom_xmat[g], m->vert_pos + 3 * (m->cloud_vertadr[cl] + v));
          for (int k = 0; k < 3; k++) { wv[k] += o->geom_xpos[g][k]; lc[k] = wv[k] - m->hfield_pos[k]; }
          HfTri t;
          if (!hf_triangle_under(m, lc[0], lc[1], &t)) continue;
          const double pa[3] = {lc[0] - t.a[0], lc[1] - t.a[1], lc[2] - t.a[2]}, dist = dot3(pa, t.n) - r;
          if (dist < best) { second = best; best = dist; memcpy(bn, t.n, sizeof bn); memcpy(bv, wv, sizeof bv); }
          else if (dist < second) second = dist;
        }
      if (best >= margin) continue;
      Contact* c = &o->contact[o->ncon++];
      c->geom = g; c->body = m->geom_bodyid[g]; c->geom1 = -1; c->body1 = 0; c->dist = best; c->tiegap = second - best;
      for (int k = 0; k < 3; k++) c->pos[k] = bv[k] - bn[k] * (r + 0.5 * best);
      set_frame(c, bn, NULL);
      contact_param(o, m->nbox, g, c);
    }
  /* robot self-collision: the statically filtered geom pairs (GqModelDesc::selfpair_*), capsule proxies in the body
   * frames, one contact per pair at the closest points of the two segments; normal from geom1 to geom2, contact point
   * midway between the surfaces (MuJoCo's convention for every pair routine).  Coincident axes (no normal) yield no contact. */
  for (int p = 0; p < m->nselfpair && o->ncon < NCON; p++) {
    const int g1 = m->selfpair_geom1[p], g2 = m->selfpair_geom2[p], b1 = m->geom_bodyid[g1], b2 = m->geom_bodyid[g2];
    { /* primitive pairs with a box: exact routines (sphere / capsule against box, box against box) */
      Prim P1, P2;
      prim_of_geom(o, g1, &P1); prim_of_geom(o, g2, &P2);
      if ((P1.kind == 2 && P2.kind != 0) || (P2.kind == 2 && P1.kind != 0)) {
        const double margin = fmax(m->geom_margin[g1], m->geom_margin[g2]);
        PairPt pts[4];
        int np, flip = 0;
        if (P1.kind == 2 && P2.kind == 2) np = box_box(P1.c, P1.R, P1.h, P2.c, P2.R, P2.h, margin, pts);
        else if (P1.kind == 2) np = capsule_box(P2.p0, P2.p1, P2.r, P1.c, P1.R, P1.h, margin, pts); /* normal box (1) -> capsule (2) */
        else { np = capsule_box(P1.p0, P1.p1, P1.r, P2.c, P2.R, P2.h, margin, pts); flip = 1; }       /* box is geom 2: flip to 1 -> 2 */
        for (int q = 0; q < np && o->ncon < NCON; q++) {
          Contact* c = &o->contact[o->ncon++];
          c->geom = g2; c->body = b2; c->geom1 = g1; c->body1 = b1; c->dist = pts[q].dist; c->tiegap = pts[q].tie;
          memcpy(c->pos, pts[q].pos, sizeof c->pos);
          double nrm[3] = {flip ? -pts[q].nrm[0] : pts[q].nrm[0], flip ? -pts[q].nrm[1] : pts[q].nrm[1], flip ? -pts[q].nrm[2] : pts[q].nrm[2]};
          set_frame(c, nrm, NULL);
          contact_param_pair(o, g1, g2, c);
        }
        continue;
      }
    }
    {
      Prim P1, P2;
      prim_of_geom(o, g1, &P1); prim_of_geom(o, g2, &P2);
      if ((P1.kind != 1 || P2.kind != 1) && m->self_convex) { /* a hull / cylinder (or a box against one): mjc_Convex on the two clouds, one contact - gq_convex.h (self_convex = 0: the capsule proxies below) */
        const int c1 = m->geom_cloudid[g1], c2 = m->geom_cloudid[g2];
        const double margin = fmax(m->geom_margin[g1], m->geom_margin[g2]);
        { /* bounding spheres (mj_collision's broad phase) */
          double dc[3] = {o->geom_xpos[g2][0] - o->geom_xpos[g1][0], o->geom_xpos[g2][1] - o->geom_xpos[g1][1], o->geom_xpos[g2][2] - o->geom_xpos[g1][2]};
          if (sqrt(dot3(dc, dc)) > m->geom_rbound[g1] + m->geom_rbound[g2] + margin) continue;
        }
        double hint[3];
        { /* mid phase: the clouds' oriented boxes */
          const double* b1x = cloud_box(o, c1); const double* b2x = cloud_box(o, c2);
          double w1[3], w2[3];
          mulmatvec3(w1, o->geom_xmat[g1], b1x); mulmatvec3(w2, o->geom_xmat[g2], b2x);
          for (int k = 0; k < 3; k++) { w1[k] += o->geom_xpos[g1][k]; w2[k] += o->geom_xpos[g2][k]; }
          g_cvx_stat[8] += 1;
          if (obb_apart(w1, o->geom_xmat[g1], b1x + 3, w2, o->geom_xmat[g2], b2x + 3, margin + m->cloud_radius[c1] + m->cloud_radius[c2], hint)) continue;
        }
        Cvx A, B;
        memset(&A, 0, sizeof A); memset(&B, 0, sizeof B);
        A.V = m->vert_pos + 3 * m->cloud_vertadr[c1]; A.nv = m->cloud_vertnum[c1]; A.r = m->cloud_radius[c1];
        memcpy(A.R, o->geom_xmat[g1], sizeof A.R); memcpy(A.t, o->geom_xpos[g1], sizeof A.t);
        B.V = m->vert_pos + 3 * m->cloud_vertadr[c2]; B.nv = m->cloud_vertnum[c2]; B.r = m->cloud_radius[c2];
        memcpy(B.R, o->geom_xmat[g2], sizeof B.R); memcpy(B.t, o->geom_xpos[g2], sizeof B.t);
        double dist, nrm[3], pos[3];
        (void)hint; /* (the mid phase's best axis as GJK's first direction was tried: boxes that overlap are not separated along their own axes, the hulls inside them hardly ever are) */
        const int full_before = o->ncon >= 12;
        const int hit_ = cvx_pair_counted(&A, &B, margin, &dist, nrm, pos, 1, NULL);
        if (full_before) g_cvx_stat[10] += g_cvx_last_queries; /* census: support queries spent on pairs behind the kernel's 12-contact capacity */
        if (!hit_) continue;
        Contact* c = &o->contact[o->ncon++];
        c->geom = g2; c->body = b2; c->geom1 = g1; c->body1 = b1; c->dist = dist; c->tiegap = g_cvx_capped ? 0.0 : (cvx_point_tie(&A, &B, nrm) > 0 ? 1.0 : -1.0); /* -1: depth and normal are determined, the contact POINT is not */
        memcpy(c->pos, pos, sizeof c->pos);
        set_frame(c, nrm, NULL);
        contact_param_pair(o, g1, g2, c);
        continue;
      }
    }
    const double* k1 = m->geom_capsule + 7 * g1; const double* k2 = m->geom_capsule + 7 * g2;
    double a0[3], a1[3], e0[3], e1[3], c1[3], c2[3], d[3];
    mulmatvec3(a0, o->xmat[b1], k1); mulmatvec3(a1, o->xmat[b1], k1 + 3);
    mulmatvec3(e0, o->xmat[b2], k2); mulmatvec3(e1, o->xmat[b2], k2 + 3);
    for (int k = 0; k < 3; k++) { a0[k] += o->xpos[b1][k]; a1[k] += o->xpos[b1][k]; e0[k] += o->xpos[b2][k]; e1[k] += o->xpos[b2][k]; }
    closest_seg_seg(a0, a1, e0, e1, c1, c2);
    for (int k = 0; k < 3; k++) d[k] = c2[k] - c1[k];
    const double len = sqrt(dot3(d, d)), dist = len - k1[6] - k2[6];
    const double margin = fmax(m->geom_margin[g1], m->geom_margin[g2]);
    if (dist >= margin || len < 1e-9) continue;
    Contact* c = &o->contact[o->ncon++];
    c->geom = g2; c->body = b2; c->geom1 = g1; c->body1 = b1; c->dist = dist; c->tiegap = 1.0;
    double nrm[3];
    for (int k = 0; k < 3; k++) { nrm[k] = d[k] / len; c->pos[k] = c1[k] + nrm[k] * (k1[6] + 0.5 * dist); }
    set_frame(c, nrm, NULL);
    contact_param_pair(o, g1, g2, c);
  }
}

/* ------------------------------------------------------------------ constraint construction */
static void get_impedance(const double* solimp, double pos, double margin, double* imp) {
  double dmin = solimp[0], dmax = solimp[1], width = solimp[2], mid = solimp[3], power = solimp[4];
  dmin = fmin(fmax(dmin, MINIMP), MAXIMP); dmax = fmin(fmax(dmax, MINIMP), MAXIMP);
  width = fmax(MINVAL, width); mid = fmin(fmax(mid, MINIMP), MAXIMP); power = fmax(1, power);
  if (dmin == dmax || width <= MINVAL) { *imp = 0.5 * (dmin + dmax); return; }
  double x = fabs(pos - margin) / width;
  if (x >= 1) { *imp = dmax; return; }
  if (x <= 0) { *imp = dmin; return; }
  double y;
  if (power == 1) y = x;
  else if (x <= mid) y = pow(x, power) / pow(mid, power - 1);
  else y = 1 - pow(1 - x, power) / pow(1 - mid, power - 1);
  *imp = dmin + y * (dmax - dmin);
}

static int add_row(GqOracle* o, int type, int id, double pos, double margin, double floss, const double* solref,
                   const double* solimp, double diag) {
  int i = o->nefc++;
  o->efc_type[i] = type; o->efc_id[i] = id; o->efc_pos[i] = pos; o->efc_margin[i] = margin;
  o->efc_frictionloss[i] = floss; o->efc_diagApprox[i] = diag;
  memcpy(o->efc_solref[i], solref, sizeof(double) * 2);
  memcpy(o->efc_solimp[i], solimp, sizeof(double) * 5);
  memset(o->efc_J[i], 0, sizeof o->efc_J[i]);
  return i;
}

static void gqo_make_constraint(GqOracle* o) {
  const GqModelDesc* m = &o->d;
  o->nefc = 0;
  /* 1. dof friction loss (mj_instantiateFriction) */
  for (int i = 0; i < m->nv; i++)
    if (m->dof_frictionloss[i] > 0) {
      int r = add_row(o, EFC_FRICTION_DOF, i, 0, 0, m->dof_frictionloss[i], m->dof_solref + 2 * i, m->dof_solimp + 5 * i,
                      m->dof_invweight0[i]);
      o->efc_J[r][i] = 1;
    }
  /* 2. joint limits (mj_instantiateLimit, hinge) */
  for (int j = 0; j < m->njnt; j++)
    if (m->jnt_limited[j] && m->jnt_type[j] == 3) {
      double value = o->qpos[m->jnt_qposadr[j]], margin = m->jnt_margin[j];
      int d = m->jnt_dofadr[j];
      for (int side = -1; side <= 1; side += 2) {
        double dist = side * (m->jnt_range[2 * j + (side + 1) / 2] - value);
        if (dist < margin) {
          int r = add_row(o, EFC_LIMIT_JOINT, j, dist, margin, 0, m->jnt_solref + 2 * j, m->jnt_solimp + 5 * j,
                          m->dof_invweight0[d]);
          o->efc_J[r][d] = -side;
        }
      }
    }
  /* 3. contacts (mj_instantiateContact), pyramidal cone */
  for (int c = 0; c < o->ncon; c++) {
    Contact* con = &o->contact[c];
    double jp[3][NV], jr[3][NV], Jc[6][NV];
    gqo_jac(o, jp, jr, con->pos, con->body);
    if (con->body1 > 0) { /* contact between two bodies of the robot: J(geom2's body) - J(geom1's body) at the contact point */
      double jp1[3][NV], jr1[3][NV];
      gqo_jac(o, jp1, jr1, con->pos, con->body1);
      for (int k = 0; k < 3; k++)
        for (int i = 0; i < NV; i++) { jp[k][i] -= jp1[k][i]; jr[k][i] -= jr1[k][i]; }
    }
    for (int k = 0; k < 3; k++)
      for (int i = 0; i < NV; i++) {
        Jc[k][i] = con->frame[3 * k] * jp[0][i] + con->frame[3 * k + 1] * jp[1][i] + con->frame[3 * k + 2] * jp[2][i];
        Jc[3 + k][i] = con->frame[3 * k] * jr[0][i] + con->frame[3 * k + 1] * jr[1][i] + con->frame[3 * k + 2] * jr[2][i];
      }
    double tran = m->body_invweight0[2 * con->body], rot = m->body_invweight0[2 * con->body + 1];
    if (con->body1 > 0) { tran += m->body_invweight0[2 * con->body1]; rot += m->body_invweight0[2 * con->body1 + 1]; }
    con->efc_address = o->nefc;
    con->mu = con->friction[0] / sqrt(m->impratio);
    if (con->dim == 1) {
      int r = add_row(o, EFC_CONTACT_FRICTIONLESS, c, con->dist, con->includemargin, 0, con->solref, con->solimp, tran);
      memcpy(o->efc_J[r], Jc[0], sizeof Jc[0]);
    } else if (m->cone == 1) { /* elliptic: one row per contact-space dimension [n, t1, t2, torsion, roll1, roll2]; only
                                 * the normal row carries the penetration (pos, margin), friction rows have pos = 0 */
      for (int k = 0; k < con->dim; k++) {
        int r = add_row(o, EFC_CONTACT_ELLIPTIC, c, k == 0 ? con->dist : 0, k == 0 ? con->includemargin : 0, 0, con->solref,
                        con->solimp, k < 3 ? tran : rot);
        memcpy(o->efc_J[r], Jc[k], sizeof Jc[k]);
      }
    } else { /* pyramidal: 2*(dim-1) edges  Jn +- mu_k * Jt_k */
      for (int k = 1; k < con->dim; k++)
        for (int s = 0; s < 2; s++) {
          int e = 2 * (k - 1) + s;
          double fri = con->friction[e / 2];
          double diag = tran + fri * fri * (e < 4 ? tran : rot);
          int r = add_row(o, EFC_CONTACT_PYRAMIDAL, c, con->dist, con->includemargin, 0, con->solref, con->solimp, diag);
          for (int i = 0; i < NV; i++) o->efc_J[r][i] = Jc[0][i] + (s ? -1.0 : 1.0) * con->friction[k - 1] * Jc[k][i];
        }
    }
  }
  /* mj_makeImpedance: R, D, aref */
  double timestep = m->timestep;
  for (int i = 0; i < o->nefc; i++) {
    double imp;
    get_impedance(o->efc_solimp[i], o->efc_pos[i], o->efc_margin[i], &imp);
    o->efc_R[i] = fmax(MINVAL, (1 - imp) * o->efc_diagApprox[i] / imp);
    double dmax = fmin(fmax(o->efc_solimp[i][1], MINIMP), MAXIMP), K, B;
    const double* sr = o->efc_solref[i];
    if (sr[0] > 0) {
      double tc = fmax(sr[0], 2 * timestep), dr = sr[1];
      K = 1 / fmax(MINVAL, dmax * dmax * tc * tc * dr * dr);
      B = 2 / fmax(MINVAL, dmax * tc);
    } else { K = -sr[0] / fmax(MINVAL, dmax * dmax); B = -sr[1] / fmax(MINVAL, dmax); }
    double vel = 0;
    for (int k = 0; k < NV; k++) vel += o->efc_J[i][k] * o->qvel[k];
    o->efc_vel[i] = vel;
    o->efc_aref[i] = -B * vel - K * imp * (o->efc_pos[i] - o->efc_margin[i]);
  }
  /* friction-adjusted R of pyramid edges: Rpy = 2 mu^2 R(first edge) for every edge of the contact */
  for (int c = 0; c < o->ncon; c++) {
    Contact* con = &o->contact[c];
    if (con->dim > 1 && m->cone == 1) { /* elliptic: R_j = R_n mu^2 / friction_j^2, so that in the scaled space
                                          * (f_n/mu, f_j/friction_j) the regulariser is isotropic and the cone circular */
      int a = con->efc_address;
      for (int j = 1; j < con->dim; j++)
        o->efc_R[a + j] = fmax(MINVAL, o->efc_R[a] * con->mu * con->mu / (con->friction[j - 1] * con->friction[j - 1]));
    } else if (con->dim > 1) {
      int a = con->efc_address;
      double Rpy = 2 * con->mu * con->mu * o->efc_R[a];
      for (int e = 0; e < 2 * (con->dim - 1); e++) o->efc_R[a + e] = fmax(MINVAL, Rpy);
    }
  }
  for (int i = 0; i < o->nefc; i++) o->efc_D[i] = 1 / o->efc_R[i];
}

/* ------------------------------------------------------------------ velocity / actuation stages */
static void gqo_com_vel(GqOracle* o) {
  const GqModelDesc* m = &o->d;
  memset(o->cvel[0], 0, sizeof o->cvel[0]);
  for (int b = 1; b < m->nbody; b++) {
    double v[6];
    memcpy(v, o->cvel[m->body_parentid[b]], sizeof v);
    for (int j = m->body_jntadr[b]; j < m->body_jntadr[b] + m->body_jntnum[b]; j++) {
      int d = m->jnt_dofadr[j];
      if (m->jnt_type[j] == 0) {
        for (int k = 0; k < 3; k++) { /* translations: world-fixed axes, cdof_dot = 0 */
          memset(o->cdof_dot[d + k], 0, sizeof o->cdof_dot[0]);
          for (int c = 0; c < 6; c++) v[c] += o->cdof[d + k][c] * o->qvel[d + k];
        }
        for (int k = 3; k < 6; k++) cross_motion(o->cdof_dot[d + k], v, o->cdof[d + k]);
        for (int k = 3; k < 6; k++)
          for (int c = 0; c < 6; c++) v[c] += o->cdof[d + k][c] * o->qvel[d + k];
      } else {
        cross_motion(o->cdof_dot[d], v, o->cdof[d]);
        for (int c = 0; c < 6; c++) v[c] += o->cdof[d][c] * o->qvel[d];
      }
    }
    memcpy(o->cvel[b], v, sizeof v);
  }
}

/* mj_rne with flg_acc = 0: Coriolis + centrifugal + gravity */
static void gqo_rne(GqOracle* o, double* res) {
  const GqModelDesc* m = &o->d;
  memset(o->cacc[0], 0, sizeof o->cacc[0]);
  for (int k = 0; k < 3; k++) o->cacc[0][3 + k] = -m->gravity[k];
  memset(o->cfrc[0], 0, sizeof o->cfrc[0]);
  for (int b = 1; b < m->nbody; b++) {
    double a[6], t1[6], t2[6];
    memcpy(a, o->cacc[m->body_parentid[b]], sizeof a);
    for (int j = m->body_jntadr[b]; j < m->body_jntadr[b] + m->body_jntnum[b]; j++) {
      int d = m->jnt_dofadr[j], n = m->jnt_type[j] == 0 ? 6 : 1;
      for (int k = 0; k < n; k++)
        for (int c = 0; c < 6; c++) a[c] += o->cdof_dot[d + k][c] * o->qvel[d + k];
    }
    memcpy(o->cacc[b], a, sizeof a);
    mul_inert_vec(t1, o->cinert[b], a);
    mul_inert_vec(t2, o->cinert[b], o->cvel[b]);
    cross_force(o->cfrc[b], o->cvel[b], t2);
    for (int c = 0; c < 6; c++) o->cfrc[b][c] += t1[c];
  }
  for (int b = m->nbody - 1; b > 0; b--) {
    int p = m->body_parentid[b];
    if (p > 0)
      for (int c = 0; c < 6; c++) o->cfrc[p][c] += o->cfrc[b][c];
  }
  for (int i = 0; i < m->nv; i++) {
    double s = 0;
    for (int c = 0; c < 6; c++) s += o->cdof[i][c] * o->cfrc[m->dof_bodyid[i]][c];
    res[i] = s;
  }
}

static void gqo_fwd_position(GqOracle* o) {
  gqo_kinematics(o);
  gqo_com_pos(o);
  gqo_crb(o);
  gqo_factor_m(o);
  gqo_collision(o);
  gqo_make_constraint(o);
}

static void gqo_fwd_velocity(GqOracle* o) {
  gqo_com_vel(o);
  for (int i = 0; i < NV; i++) o->qfrc_passive[i] = -o->d.dof_damping[i] * o->qvel[i];
  gqo_rne(o, o->qfrc_bias);
  /* efc_vel / aref depend on qvel: recompute (make_constraint above already used the current qvel) */
}

static void gqo_fwd_actuation(GqOracle* o) {
  const GqModelDesc* m = &o->d;
  memset(o->qfrc_actuator, 0, sizeof o->qfrc_actuator);
  for (int u = 0; u < m->nu; u++) {
    double c = o->ctrl[u];
    if (m->actuator_ctrllimited[u]) c = fmin(fmax(c, m->actuator_ctrlrange[2 * u]), m->actuator_ctrlrange[2 * u + 1]);
    double f = c; /* motor: gain 1, no bias */
    if (m->actuator_forcelimited[u]) f = fmin(fmax(f, m->actuator_forcerange[2 * u]), m->actuator_forcerange[2 * u + 1]);
    o->qfrc_actuator[m->jnt_dofadr[m->actuator_trnid[u]]] += m->actuator_gear[u] * f;
  }
  for (int j = 0; j < m->njnt; j++)
    if (m->jnt_actfrclimited[j] && m->jnt_type[j] == 3) {
      int d = m->jnt_dofadr[j];
      o->qfrc_actuator[d] = fmin(fmax(o->qfrc_actuator[d], m->jnt_actfrcrange[2 * j]), m->jnt_actfrcrange[2 * j + 1]);
    }
}

static void gqo_fwd_acceleration(GqOracle* o) {
  for (int i = 0; i < NV; i++)
    o->qfrc_smooth[i] = o->qfrc_passive[i] - o->qfrc_bias[i] + o->qfrc_actuator[i] + o->qfrc_applied[i];
  memcpy(o->qacc_smooth, o->qfrc_smooth, sizeof o->qacc_smooth);
  gqo_solve_m(o, o->qacc_smooth);
}

/* ------------------------------------------------------------------ constraint force law (mj_constraintUpdate):
 * force and cost of every row given jar = J*qacc - aref.  Returns the constraint cost. */
/* Elliptic contact at residual z = jar[a..a+dim) (mj_constraintUpdate, mjCNSTR_CONTACT_ELLIPTIC).  With
 * N = mu z_0, U_j = friction_j z_j, T = |U|:  top zone (N >= mu T): cost 0;  bottom zone (mu N + T <= 0): plain
 * quadratic sum_j D_j z_j^2 / 2;  middle zone: Dm (N - mu T)^2 / 2 with Dm = D_0 / (mu^2 (1 + mu^2)) - the dual of
 * projecting onto the (scaled, circular) friction cone.  Returns the zone (0 top, 1 bottom, 2 middle); grad = ds/dz;
 * hess (dim x dim, row-major, may be NULL) = d2s/dz2. */
static int elliptic_eval(const GqOracle* o, const Contact* con, const double* z, double* cost, double* grad, double* hess) {
  const int dim = con->dim, a = con->efc_address;
  const double mu = con->mu, *fri = con->friction;
  double U[6], T = 0;
  const double N = mu * z[0];
  for (int j = 1; j < dim; j++) { U[j] = fri[j - 1] * z[j]; T += U[j] * U[j]; }
  T = sqrt(T);
  *cost = 0;
  for (int j = 0; j < dim; j++) grad[j] = 0;
  if (hess) memset(hess, 0, sizeof(double) * 36);
  if (N >= mu * T || (T <= 0 && N >= 0)) return 0;
  if (mu * N + T <= 0 || (T <= 0 && N < 0)) {
    for (int j = 0; j < dim; j++) {
      const double D = o->efc_D[a + j];
      *cost += 0.5 * D * z[j] * z[j]; grad[j] = D * z[j];
      if (hess) hess[6 * j + j] = D;
    }
    return 1;
  }
  const double Dm = o->efc_D[a] / (mu * mu * (1 + mu * mu)), q = N - mu * T;
  *cost = 0.5 * Dm * q * q;
  double g[6];
  g[0] = mu;
  for (int j = 1; j < dim; j++) g[j] = -mu * fri[j - 1] * U[j] / T;
  for (int j = 0; j < dim; j++) grad[j] = Dm * q * g[j];
  if (hess) {
    for (int i = 0; i < dim; i++)
      for (int j = 0; j < dim; j++) hess[6 * i + j] = Dm * g[i] * g[j];
    /* q * d2q/dz2, d2q = -mu d2T:  d2T_ij = f_i f_j (delta_ij - u_i u_j) / T */
    for (int i = 1; i < dim; i++)
      for (int j = 1; j < dim; j++) {
        const double ui = U[i] / T, uj = U[j] / T;
        hess[6 * i + j] += Dm * q * (-mu) * fri[i - 1] * fri[j - 1] * ((i == j ? 1.0 : 0.0) - ui * uj) / T;
      }
  }
  return 2;
}

static double constraint_update(GqOracle* o, const double* jar, double* force, int* active) {
  double cost = 0;
  for (int i = 0; i < o->nefc; i++) {
    double R = o->efc_R[i], D = o->efc_D[i], x = jar[i];
    int act = 1;
    if (o->efc_type[i] == EFC_CONTACT_ELLIPTIC) { /* whole contact at once; active[] holds the zone on every row */
      const Contact* con = &o->contact[o->efc_id[i]];
      double c, g[6];
      const int zone = elliptic_eval(o, con, jar + i, &c, g, NULL);
      cost += c;
      for (int j = 0; j < con->dim; j++) { force[i + j] = -g[j]; if (active) active[i + j] = zone; }
      i += con->dim - 1;
      continue;
    }
    if (o->efc_type[i] == EFC_FRICTION_DOF) {
      double f = o->efc_frictionloss[i];
      if (x <= -R * f) { force[i] = f; cost += -0.5 * R * f * f - f * x; act = 0; }
      else if (x >= R * f) { force[i] = -f; cost += -0.5 * R * f * f + f * x; act = 0; }
      else { force[i] = -D * x; cost += 0.5 * D * x * x; }
    } else { /* limit, frictionless / pyramidal contact: one-sided quadratic */
      if (x < 0) { force[i] = -D * x; cost += 0.5 * D * x * x; }
      else { force[i] = 0; act = 0; }
    }
    if (active) active[i] = act;
  }
  return cost;
}

/* ------------------------------------------------------------------ mj_solNewton (primal, exact line search) */
typedef struct { double a; double c0, c1, c2; } Breakpt; /* at alpha >= a add (c0,c1,c2) to the quadratic */
static int cmp_bp(const void* x, const void* y) {
  double a = ((const Breakpt*)x)->a, b = ((const Breakpt*)y)->a;
  return a < b ? -1 : a > b;
}

static double primal_cost(GqOracle* o, const double* qacc, double* force) {
  double jar[NEFC], Ma[NV], cost = 0;
  for (int i = 0; i < o->nefc; i++) {
    double s = -o->efc_aref[i];
    for (int k = 0; k < NV; k++) s += o->efc_J[i][k] * qacc[k];
    jar[i] = s;
  }
  for (int i = 0; i < NV; i++) {
    double s = 0;
    for (int k = 0; k < NV; k++) s += o->M[i][k] * (qacc[k] - o->qacc_smooth[k]);
    Ma[i] = s;
  }
  for (int i = 0; i < NV; i++) cost += 0.5 * Ma[i] * (qacc[i] - o->qacc_smooth[i]);
  double tmp[NEFC];
  return cost + constraint_update(o, jar, force ? force : tmp, NULL);
}

/* phi'(alpha), phi''(alpha) of the total cost along the search direction (jar + alpha*jv per row) */
static void line_derivs(const GqOracle* o, const double* jar, const double* jv, double alpha, double q1, double q2,
                        double* d1, double* d2) {
  double s1 = q1 + 2 * q2 * alpha, s2 = 2 * q2;
  for (int r = 0; r < o->nefc; r++) {
    const double x = jar[r] + alpha * jv[r], v = jv[r], D = o->efc_D[r], R = o->efc_R[r];
    if (o->efc_type[r] == EFC_CONTACT_ELLIPTIC) {
      const Contact* con = &o->contact[o->efc_id[r]];
      double z[6], c, g[6], hs[36];
      for (int j = 0; j < con->dim; j++) z[j] = jar[r + j] + alpha * jv[r + j];
      elliptic_eval(o, con, z, &c, g, hs);
      for (int a = 0; a < con->dim; a++) {
        s1 += g[a] * jv[r + a];
        for (int b = 0; b < con->dim; b++) s2 += jv[r + a] * hs[6 * a + b] * jv[r + b];
      }
      r += con->dim - 1;
    } else if (o->efc_type[r] == EFC_FRICTION_DOF) {
      const double f = o->efc_frictionloss[r];
      if (x <= -R * f) s1 += -f * v;
      else if (x >= R * f) s1 += f * v;
      else { s1 += D * x * v; s2 += D * v * v; }
    } else if (x < 0) { s1 += D * x * v; s2 += D * v * v; }
  }
  *d1 = s1; *d2 = s2;
}

static void gqo_sol_newton(GqOracle* o, int maxiter, double tol) {
  const int nefc = o->nefc;
  double qacc[NV], jar[NEFC], grad[NV], search[NV], jv[NEFC], Mv[NV], H[NV][NV];
  int active[NEFC];
  /* warm start: whichever of qacc_warmstart / qacc_smooth has the lower cost */
  double cw = primal_cost(o, o->qacc_warmstart, NULL), cs = primal_cost(o, o->qacc_smooth, NULL);
  memcpy(qacc, cw < cs ? o->qacc_warmstart : o->qacc_smooth, sizeof qacc);
  { /* experiment knob (tools/newton_start_experiment.py): GQO_NEWTON_START = 1 qacc_smooth, 2 zero, 3 warm start */
    static int mode = -1;
    if (mode < 0) { const char* e = getenv("GQO_NEWTON_START"); mode = e ? atoi(e) : 0; }
    if (mode == 1) memcpy(qacc, o->qacc_smooth, sizeof qacc);
    else if (mode == 2) memset(qacc, 0, sizeof qacc);
    else if (mode == 3) memcpy(qacc, o->qacc_warmstart, sizeof qacc);
  }
  double scale = 1.0 / (o->d.meaninertia * NV);
  int iter = 0;
  for (; iter < maxiter; iter++) {
    for (int i = 0; i < nefc; i++) {
      double s = -o->efc_aref[i];
      for (int k = 0; k < NV; k++) s += o->efc_J[i][k] * qacc[k];
      jar[i] = s;
    }
    constraint_update(o, jar, o->efc_force, active);
    /* gradient = M (qacc - qacc_smooth) - J' f */
    for (int i = 0; i < NV; i++) {
      double s = 0;
      for (int k = 0; k < NV; k++) s += o->M[i][k] * (qacc[k] - o->qacc_smooth[k]);
      for (int r = 0; r < nefc; r++) s -= o->efc_J[r][i] * o->efc_force[r];
      grad[i] = s;
    }
    double gn = 0;
    for (int i = 0; i < NV; i++) gn += grad[i] * grad[i];
    if (scale * sqrt(gn) < tol) break;
    /* Hessian = M + J' diag(D active) J ; dense Cholesky */
    for (int i = 0; i < NV; i++)
      for (int j = 0; j < NV; j++) H[i][j] = o->M[i][j];
    int has_cone = 0;
    for (int r = 0; r < nefc; r++) {
      if (o->efc_type[r] == EFC_CONTACT_ELLIPTIC) { /* dense dim x dim block: H += Jc' (d2s/dz2) Jc */
        const Contact* con = &o->contact[o->efc_id[r]];
        double c, g[6], hs[36];
        const int zone = elliptic_eval(o, con, jar + r, &c, g, hs);
        has_cone = 1;
        if (zone != 0)
          for (int a = 0; a < con->dim; a++)
            for (int b = 0; b < con->dim; b++) {
              const double w = hs[6 * a + b];
              if (w != 0)
                for (int i = 0; i < NV; i++) {
                  const double t = o->efc_J[r + a][i] * w;
                  if (t != 0)
                    for (int j = 0; j < NV; j++) H[i][j] += t * o->efc_J[r + b][j];
                }
            }
        r += con->dim - 1;
        continue;
      }
      if (active[r])
        for (int i = 0; i < NV; i++) {
          double a = o->efc_J[r][i] * o->efc_D[r];
          if (a != 0)
            for (int j = 0; j < NV; j++) H[i][j] += a * o->efc_J[r][j];
        }
    }
    for (int j = 0; j < NV; j++) {
      for (int k = 0; k < j; k++)
        for (int i = j; i < NV; i++) H[i][j] -= H[i][k] * H[j][k];
      double dj = sqrt(fmax(H[j][j], MINVAL));
      for (int i = j; i < NV; i++) H[i][j] /= dj;
    }
    for (int i = 0; i < NV; i++) search[i] = -grad[i];
    for (int i = 0; i < NV; i++) {
      for (int k = 0; k < i; k++) search[i] -= H[i][k] * search[k];
      search[i] /= H[i][i];
    }
    for (int i = NV - 1; i >= 0; i--) {
      for (int k = i + 1; k < NV; k++) search[i] -= H[k][i] * search[k];
      search[i] /= H[i][i];
    }
    /* exact line search on phi(alpha) = cost(qacc + alpha*search): convex piecewise quadratic.
     * Gauss part: q0 + q1 a + q2 a^2 with q1 = search.(M(qacc-qs)), q2 = 0.5 search.M.search */
    for (int i = 0; i < NV; i++) {
      double s = 0;
      for (int k = 0; k < NV; k++) s += o->M[i][k] * search[k];
      Mv[i] = s;
    }
    for (int r = 0; r < nefc; r++) {
      double s = 0;
      for (int k = 0; k < NV; k++) s += o->efc_J[r][k] * search[k];
      jv[r] = s;
    }
    double q1 = 0, q2 = 0;
    for (int i = 0; i < NV; i++) {
      double s = 0;
      for (int k = 0; k < NV; k++) s += o->M[i][k] * (qacc[k] - o->qacc_smooth[k]);
      q1 += search[i] * s;
      q2 += 0.5 * search[i] * Mv[i];
    }
    if (has_cone) { /* elliptic contacts: phi is convex and C1 but not piecewise quadratic -> safeguarded Newton on phi' */
      { /* experiment knob GQO_LS_MODE = 1 (see below): the full Newton step whenever it lowers the cost */
        static int lsmode_c = -1;
        static long n_full = 0, n_fall = 0;
        if (lsmode_c < 0) { const char* e = getenv("GQO_LS_MODE"); lsmode_c = e ? atoi(e) : 0; }
        if (lsmode_c == 1) {
          double trial[NV];
          for (int i = 0; i < NV; i++) trial[i] = qacc[i] + search[i];
          const int okf = primal_cost(o, trial, NULL) < primal_cost(o, qacc, NULL);
          if (okf) n_full++; else n_fall++;
          if (getenv("GQO_LS_STATS") && ((n_full + n_fall) % 5000) == 0) fprintf(stderr, "ls mode 1 (cone): %ld full steps accepted, %ld fell back to the search\n", n_full, n_fall);
          if (okf) { memcpy(qacc, trial, sizeof qacc); continue; }
        }
      }
      double lo = 0, hi = -1, alpha = 0, g0 = 0;
      int ok = 0;
      for (int it = 0; it < 200; it++) {
        double d1, d2;
        line_derivs(o, jar, jv, alpha, q1, q2, &d1, &d2);
        if (it == 0) {
          g0 = d1;
          if (!(g0 < 0)) break; /* not a descent direction: converged */
          ok = 1;
          alpha = d2 > 0 ? -d1 / d2 : 1.0;
          continue;
        }
        if (fabs(d1) <= 1e-13 * fabs(g0)) break;
        if (d1 < 0) lo = alpha; else hi = alpha;
        double an = d2 > 0 ? alpha - d1 / d2 : -1;
        if (!(an > lo) || (hi > 0 && !(an < hi))) an = hi > 0 ? 0.5 * (lo + hi) : 2 * alpha;
        if (hi > 0 && hi - lo <= 1e-15 * hi) { alpha = an; break; }
        alpha = an;
      }
      if (!ok || alpha <= 0) break;
      for (int i = 0; i < NV; i++) qacc[i] += alpha * search[i];
      continue;
    }
    /* derivative phi'(a) = d1 + 2*d2*a, accumulate row pieces valid at a = 0+, then sweep breakpoints */
    static Breakpt bp[2 * NEFC];
    int nbp = 0;
    double d1 = q1, d2 = q2;
    for (int r = 0; r < nefc; r++) {
      double x = jar[r], v = jv[r], D = o->efc_D[r], R = o->efc_R[r];
      if (v == 0) { /* constant piece: contributes nothing to the derivative */ continue; }
      if (o->efc_type[r] == EFC_FRICTION_DOF) {
        double f = o->efc_frictionloss[r], lo = -R * f, hi = R * f;
        /* pieces along alpha: x + alpha v crosses lo and hi */
        double a_lo = (lo - x) / v, a_hi = (hi - x) / v;
        double a1 = fmin(a_lo, a_hi), a2 = fmax(a_lo, a_hi);
        /* zone slopes: below lo: cost' = -f v ; quad: D (x+av) v ; above hi: +f v */
        double s_first = v > 0 ? -f * v : f * v; /* zone entered from alpha -> -inf */
        /* piece functions of derivative: linear zone -> (c1 = s, c2 = 0); quad zone -> (c1 = D x v, 2 c2 = D v v) */
        double quad1 = D * x * v, quad2 = 0.5 * D * v * v, s_last = -s_first;
        if (0 < a1) { d1 += s_first; bp[nbp++] = (Breakpt){a1, 0, quad1 - s_first, quad2};
                      bp[nbp++] = (Breakpt){a2, 0, s_last - quad1, -quad2}; }
        else if (0 < a2) { d1 += quad1; d2 += quad2; bp[nbp++] = (Breakpt){a2, 0, s_last - quad1, -quad2}; }
        else { d1 += s_last; }
      } else {
        double a0 = -x / v; /* x + a v = 0 */
        double quad1 = D * x * v, quad2 = 0.5 * D * v * v;
        if (v > 0) { /* active (x<0) before a0, inactive after */
          if (0 < a0) { d1 += quad1; d2 += quad2; bp[nbp++] = (Breakpt){a0, 0, -quad1, -quad2}; }
        } else {     /* inactive before a0, active after */
          if (0 < a0) bp[nbp++] = (Breakpt){a0, 0, quad1, quad2};
          else { d1 += quad1; d2 += quad2; }
        }
      }
    }
    qsort(bp, nbp, sizeof(Breakpt), cmp_bp);
    double alpha = 0;
    int found = 0;
    { /* experiment knob GQO_LS_MODE = 1: take the full Newton step whenever it lowers the cost (semi-smooth Newton) */
      static int lsmode = -1;
      if (lsmode < 0) { const char* e = getenv("GQO_LS_MODE"); lsmode = e ? atoi(e) : 0; }
      if (lsmode == 1) {
        double trial[NV];
        for (int i = 0; i < NV; i++) trial[i] = qacc[i] + search[i];
        static long n_full = 0, n_fall = 0;
        const int okf = primal_cost(o, trial, NULL) < primal_cost(o, qacc, NULL);
        if (okf) n_full++; else n_fall++;
        if (getenv("GQO_LS_STATS") && ((n_full + n_fall) % 5000) == 0) fprintf(stderr, "ls mode 1: %ld full steps accepted, %ld fell back to the exact search\n", n_full, n_fall);
        if (okf) { memcpy(qacc, trial, sizeof qacc); continue; }
      }
    }
    if (d1 >= 0) { alpha = 0; found = 1; }
    for (int k = 0; k <= nbp && !found; k++) {
      double hi = k < nbp ? bp[k].a : 1e300;
      /* root of d1 + 2 d2 a in the current segment */
      if (d2 > 0) {
        double a = -d1 / (2 * d2);
        if (a <= hi) { alpha = a; found = 1; break; }
      }
      if (k < nbp) { d1 += bp[k].c1; d2 += bp[k].c2;
        if (d1 + 2 * d2 * hi >= 0) { alpha = hi; found = 1; break; } }
    }
    if (!found || alpha <= 0) break;
    for (int i = 0; i < NV; i++) qacc[i] += alpha * search[i];
  }
  o->solver_niter = iter;
  for (int i = 0; i < nefc; i++) {
    double s = -o->efc_aref[i];
    for (int k = 0; k < NV; k++) s += o->efc_J[i][k] * qacc[k];
    jar[i] = s;
  }
  constraint_update(o, jar, o->efc_force, NULL);
  memcpy(o->qacc, qacc, sizeof qacc);
}

/* ------------------------------------------------------------------ mj_solPGS (dual) */
static double g_AR[NEFC][NEFC]; /* not re-entrant: PGS leg of the oracle is single-threaded */
static void gqo_sol_pgs(GqOracle* o, int maxiter, double tol) {
  const int nefc = o->nefc;
  static double B[NEFC][NV];
  for (int i = 0; i < nefc; i++) {
    memcpy(B[i], o->efc_J[i], sizeof B[i]);
    gqo_solve_m(o, B[i]);
  }
  for (int i = 0; i < nefc; i++)
    for (int j = 0; j < nefc; j++) {
      double s = 0;
      for (int k = 0; k < NV; k++) s += o->efc_J[i][k] * B[j][k];
      g_AR[i][j] = s + (i == j ? o->efc_R[i] : 0);
    }
  for (int i = 0; i < nefc; i++) {
    double s = -o->efc_aref[i];
    for (int k = 0; k < NV; k++) s += o->efc_J[i][k] * o->qacc_smooth[k];
    o->efc_b[i] = s;
  }
  /* warm start: forces from the primal force law at qacc_warmstart; keep only if the dual cost is negative */
  double jar[NEFC] = {0}, *f = o->efc_force;
  for (int i = 0; i < nefc; i++) {
    double s = -o->efc_aref[i];
    for (int k = 0; k < NV; k++) s += o->efc_J[i][k] * o->qacc_warmstart[k];
    jar[i] = s;
  }
  constraint_update(o, jar, f, NULL);
  double cost = 0;
  for (int i = 0; i < nefc; i++) {
    double s = 0;
    for (int j = 0; j < nefc; j++) s += g_AR[i][j] * f[j];
    cost += 0.5 * f[i] * s + f[i] * o->efc_b[i];
  }
  if (cost > 0) memset(f, 0, sizeof(double) * nefc);
  double scale = 1.0 / (o->d.meaninertia * NV);
  int iter = 0;
  for (; iter < maxiter; iter++) {
    double improvement = 0;
    for (int i = 0; i < nefc; i++) {
      double res = o->efc_b[i];
      for (int j = 0; j < nefc; j++) res += g_AR[i][j] * f[j];
      double old = f[i], x = old - res / g_AR[i][i];
      if (o->efc_type[i] == EFC_FRICTION_DOF) { double fl = o->efc_frictionloss[i]; x = fmin(fmax(x, -fl), fl); }
      else x = fmax(0, x);
      double delta = x - old;
      f[i] = x;
      improvement -= 0.5 * delta * delta * g_AR[i][i] + delta * res;
    }
    if (improvement * scale < tol) { iter++; break; }
  }
  o->solver_niter = iter;
  double qc[NV];
  memset(qc, 0, sizeof qc);
  for (int i = 0; i < nefc; i++)
    for (int k = 0; k < NV; k++) qc[k] += o->efc_J[i][k] * f[i];
  gqo_solve_m(o, qc);
  for (int k = 0; k < NV; k++) o->qacc[k] = o->qacc_smooth[k] + qc[k];
}

static void gqo_fwd_constraint(GqOracle* o) {
  if (o->nefc == 0) {
    memcpy(o->qacc, o->qacc_smooth, sizeof o->qacc);
    memset(o->qfrc_constraint, 0, sizeof o->qfrc_constraint);
    o->solver_niter = 0;
    return;
  }
  if (o->d.solver == 1 || o->d.cone == 1) gqo_sol_newton(o, o->d.iterations, o->d.tolerance); /* no elliptic PGS (needs the per-contact QCQP) */
  else gqo_sol_pgs(o, o->d.iterations, o->d.tolerance);
  memset(o->qfrc_constraint, 0, sizeof o->qfrc_constraint);
  for (int i = 0; i < o->nefc; i++)
    for (int k = 0; k < NV; k++) o->qfrc_constraint[k] += o->efc_J[i][k] * o->efc_force[i];
}

/* ------------------------------------------------------------------ mj_Euler */
static void gqo_euler(GqOracle* o) {
  const GqModelDesc* m = &o->d;
  double h = m->timestep, qacc[NV];
  int damped = 0;
  for (int i = 0; i < NV; i++) damped |= m->dof_damping[i] > 0;
  if (damped) { /* (M + h*diag(damping)) qacc' = qfrc_smooth + qfrc_constraint */
    double LD[NV][NV];
    for (int i = 0; i < NV; i++)
      for (int j = 0; j < NV; j++) LD[i][j] = j <= i ? o->M[i][j] : 0.0;
    for (int i = 0; i < NV; i++) LD[i][i] += h * m->dof_damping[i];
    factor_ld(m, LD);
    for (int i = 0; i < NV; i++) qacc[i] = o->qfrc_smooth[i] + o->qfrc_constraint[i];
    solve_ld(m, LD, qacc);
  } else memcpy(qacc, o->qacc, sizeof qacc);
  for (int i = 0; i < NV; i++) o->qvel[i] += h * qacc[i];
  /* mj_integratePos */
  for (int j = 0; j < m->njnt; j++) {
    int qa = m->jnt_qposadr[j], d = m->jnt_dofadr[j];
    if (m->jnt_type[j] == 0) {
      for (int k = 0; k < 3; k++) o->qpos[qa + k] += h * o->qvel[d + k];
      double w[3] = {o->qvel[d + 3], o->qvel[d + 4], o->qvel[d + 5]};
      double n = sqrt(dot3(w, w));
      if (n > MINVAL) {
        double ax[3] = {w[0] / n, w[1] / n, w[2] / n}, qr[4];
        axis_angle2quat(qr, ax, h * n);
        quat_mul(o->qpos + qa + 3, o->qpos + qa + 3, qr);
      }
      quat_normalize(o->qpos + qa + 3);
    } else o->qpos[qa] += h * o->qvel[d];
  }
  o->time += h;
  memcpy(o->qacc_warmstart, o->qacc, sizeof o->qacc);
}

/* ------------------------------------------------------------------ public stepping API */
static int bad(const double* x, int n, double lim) {
  for (int i = 0; i < n; i++)
    if (!(fabs(x[i]) < lim)) return 1;
  return 0;
}

/* mj_forward; stage 1 = position + velocity only (what mj_step1 evaluates, quadruped_env.py:376) */
int gqo_forward(GqOracle* o, const double* ctrl, int stage) {
  if (ctrl) memcpy(o->ctrl, ctrl, sizeof(double) * o->d.nu);
  gqo_fwd_position(o);
  gqo_fwd_velocity(o);
  if (stage == 1) return GQ_OK;
  gqo_fwd_actuation(o);
  gqo_fwd_acceleration(o);
  gqo_fwd_constraint(o);
  { /* mj_sensorVel / mj_sensorAcc for a gyro + accelerometer pair on a base-body site: site-frame angular velocity,
     * site-frame acceleration of the site point minus gravity (reads +9.81 along world z at rest) */
    const double* R = o->xmat[1];
    double Rs[9], RS[9], r[3], ww[3], aw[3], t1[3], t2[3], ap[3];
    quat2mat(Rs, o->imu_quat);
    mulmat3(RS, R, Rs);
    mulmatvec3(r, R, o->imu_pos);
    mulmatvec3(ww, R, o->qvel + 3);
    mulmatvec3(aw, R, o->qacc + 3);
    cross3(t1, aw, r); cross3(t2, ww, r); cross3(t2, ww, t2);
    for (int k = 0; k < 3; k++) ap[k] = o->qacc[k] + t1[k] + t2[k] - o->d.gravity[k];
    mulmatTvec3(o->imu_acc, RS, ap);
    mulmatTvec3(o->imu_gyro, Rs, o->qvel + 3);
  }
  return GQ_OK;
}

int gqo_set_imu(GqOracle* o, const double* pos, const double* quat) {
  memcpy(o->imu_pos, pos, sizeof o->imu_pos);
  memcpy(o->imu_quat, quat, sizeof o->imu_quat);
  quat_normalize(o->imu_quat);
  return GQ_OK;
}

int gqo_step(GqOracle* o, const double* ctrl) {
  o->warning = bad(o->qpos, NQ, 1e10) | bad(o->qvel, NV, 1e10);
  gqo_forward(o, ctrl, 0);
  o->warning |= bad(o->qacc, NV, 1e10);
  gqo_euler(o);
  return GQ_OK;
}

int gqo_set_state(GqOracle* o, const double* qpos, const double* qvel, const double* qacc_warmstart,
                  const double* qfrc_applied, double time, double friction) {
  if (qpos) memcpy(o->qpos, qpos, sizeof o->qpos);
  if (qvel) memcpy(o->qvel, qvel, sizeof o->qvel);
  if (qacc_warmstart) memcpy(o->qacc_warmstart, qacc_warmstart, sizeof o->qacc_warmstart);
  if (qfrc_applied) memcpy(o->qfrc_applied, qfrc_applied, sizeof o->qfrc_applied);
  o->time = time;
  o->friction = friction;
  return GQ_OK;
}

/* value of the primal objective mj_solNewton minimises, at an arbitrary qacc (after gqo_forward built the rows) */
double gqo_primal_cost(GqOracle* o, const double* qacc) { return primal_cost(o, qacc, NULL); }

int gqo_set_solver(GqOracle* o, int solver, int iterations, double tolerance) {
  o->d.solver = solver; o->d.iterations = iterations; o->d.tolerance = tolerance;
  return GQ_OK;
}

/* mj_contactForce: normal/tangent force of contact c in the contact frame (mju_decodePyramid) */
static void gqo_contact_force(const GqOracle* o, int c, double* out6) {
  const Contact* con = &o->contact[c];
  memset(out6, 0, sizeof(double) * 6);
  const double* f = o->efc_force + con->efc_address;
  if (con->dim == 1) { out6[0] = f[0]; return; }
  if (o->d.cone == 1) { for (int k = 0; k < con->dim; k++) out6[k] = f[k]; return; } /* elliptic: rows are the contact-space force */
  for (int e = 0; e < 2 * (con->dim - 1); e++) out6[0] += f[e];
  for (int k = 0; k < con->dim - 1; k++) out6[k + 1] = (f[2 * k] - f[2 * k + 1]) * con->friction[k];
}

#define GET(nm, ptr, count)                                  \
  if (!strcmp(name, nm)) {                                   \
    int n = (count);                                         \
    if (n > max_n) n = max_n;                                \
    memcpy(out, (ptr), sizeof(double) * (size_t)n);          \
    return n;                                                \
  }

int gqo_get(const GqOracle* o, const char* name, double* out, int max_n) {
  const GqModelDesc* m = &o->d;
  GET("qpos", o->qpos, NQ) GET("qvel", o->qvel, NV) GET("qacc", o->qacc, NV) GET("qacc_warmstart", o->qacc_warmstart, NV)
  GET("ctrl", o->ctrl, m->nu) GET("qfrc_applied", o->qfrc_applied, NV) GET("time", &o->time, 1)
  GET("xpos", o->xpos, 3 * m->nbody) GET("xquat", o->xquat, 4 * m->nbody) GET("xmat", o->xmat, 9 * m->nbody)
  GET("xipos", o->xipos, 3 * m->nbody) GET("subtree_com", o->subtree_com, 3 * m->nbody)
  GET("geom_xpos", o->geom_xpos, 3 * m->ngeom) GET("geom_xmat", o->geom_xmat, 9 * m->ngeom)
  GET("M", o->M, NV * NV) GET("qfrc_bias", o->qfrc_bias, NV) GET("qfrc_passive", o->qfrc_passive, NV)
  GET("qfrc_actuator", o->qfrc_actuator, NV) GET("qfrc_smooth", o->qfrc_smooth, NV) GET("qacc_smooth", o->qacc_smooth, NV)
  GET("imu_acc", o->imu_acc, 3) GET("imu_gyro", o->imu_gyro, 3)
  GET("qfrc_constraint", o->qfrc_constraint, NV) GET("cvel", o->cvel, 6 * m->nbody)
  GET("efc_pos", o->efc_pos, o->nefc) GET("efc_margin", o->efc_margin, o->nefc) GET("efc_R", o->efc_R, o->nefc)
  GET("efc_D", o->efc_D, o->nefc) GET("efc_aref", o->efc_aref, o->nefc) GET("efc_vel", o->efc_vel, o->nefc)
  GET("efc_force", o->efc_force, o->nefc) GET("efc_frictionloss", o->efc_frictionloss, o->nefc)
  GET("efc_diagApprox", o->efc_diagApprox, o->nefc) GET("efc_b", o->efc_b, o->nefc)
  if (!strcmp(name, "efc_J")) {
    int n = o->nefc * NV; if (n > max_n) n = max_n;
    for (int i = 0; i < n; i++) out[i] = o->efc_J[i / NV][i % NV];
    return n;
  }
  if (!strcmp(name, "efc_type")) { int n = o->nefc < max_n ? o->nefc : max_n; for (int i = 0; i < n; i++) out[i] = o->efc_type[i]; return n; }
  if (!strcmp(name, "nefc")) { out[0] = o->nefc; return 1; }
  if (!strcmp(name, "ncon")) { out[0] = o->ncon; return 1; }
  if (!strcmp(name, "solver_niter")) { out[0] = o->solver_niter; return 1; }
  if (!strcmp(name, "warning")) { out[0] = o->warning; return 1; }
  if (!strcmp(name, "contact_dist")) { int n = o->ncon < max_n ? o->ncon : max_n; for (int i = 0; i < n; i++) out[i] = o->contact[i].dist; return n; }
  if (!strcmp(name, "contact_tiegap")) { int n = o->ncon < max_n ? o->ncon : max_n; for (int i = 0; i < n; i++) out[i] = o->contact[i].tiegap; return n; }
  if (!strcmp(name, "contact_geom")) { int n = o->ncon < max_n ? o->ncon : max_n; for (int i = 0; i < n; i++) out[i] = o->contact[i].geom; return n; }
  if (!strcmp(name, "contact_dim")) { int n = o->ncon < max_n ? o->ncon : max_n; for (int i = 0; i < n; i++) out[i] = o->contact[i].dim; return n; }
  if (!strcmp(name, "contact_mu")) { int n = o->ncon < max_n ? o->ncon : max_n; for (int i = 0; i < n; i++) out[i] = o->contact[i].mu; return n; }
  if (!strcmp(name, "contact_efc_address")) { int n = o->ncon < max_n ? o->ncon : max_n; for (int i = 0; i < n; i++) out[i] = o->contact[i].efc_address; return n; }
  if (!strcmp(name, "contact_friction")) { int n = 5 * o->ncon < max_n ? 5 * o->ncon : max_n; for (int i = 0; i < n; i++) out[i] = o->contact[i / 5].friction[i % 5]; return n; }
  if (!strcmp(name, "contact_body1")) { int n = o->ncon < max_n ? o->ncon : max_n; for (int i = 0; i < n; i++) out[i] = o->contact[i].body1; return n; }
  if (!strcmp(name, "contact_geom1")) { int n = o->ncon < max_n ? o->ncon : max_n; for (int i = 0; i < n; i++) out[i] = o->contact[i].geom1; return n; }
  if (!strcmp(name, "contact_body")) { int n = o->ncon < max_n ? o->ncon : max_n; for (int i = 0; i < n; i++) out[i] = o->contact[i].body; return n; }
  if (!strcmp(name, "contact_pos")) { int n = 3 * o->ncon < max_n ? 3 * o->ncon : max_n; for (int i = 0; i < n; i++) out[i] = o->contact[i / 3].pos[i % 3]; return n; }
  if (!strcmp(name, "contact_frame")) { int n = 9 * o->ncon < max_n ? 9 * o->ncon : max_n; for (int i = 0; i < n; i++) out[i] = o->contact[i / 9].frame[i % 9]; return n; }
  if (!strcmp(name, "contact_force")) { /* mj_contactForce for every contact, 6 each */
    int n = 6 * o->ncon < max_n ? 6 * o->ncon : max_n; double f[6];
    for (int c = 0; c * 6 < n; c++) { gqo_contact_force(o, c, f); for (int k = 0; k < 6 && 6 * c + k < n; k++) out[6 * c + k] = f[k]; }
    return n;
  }
  snprintf(g_err, sizeof g_err, "gqo_get: unknown field %s", name);
  return GQ_EINVAL;
}

/* mj_jac wrapper for tests / observation oracle: point in world coords, body id; jacp/jacr 3 x nv row-major */
int gqo_jac_point(const GqOracle* o, const double* point, int body, double* jacp, double* jacr) {
  double jp[3][NV], jr[3][NV];
  gqo_jac(o, jp, jr, point, body);
  if (jacp) memcpy(jacp, jp, sizeof jp);
  if (jacr) memcpy(jacr, jr, sizeof jr);
  return GQ_OK;
}

/* ------------------------------------------------------------------ observation + termination assembly in C
 * (restates quadruped_env.py:1146-1257 on top of the oracle's mjData-equivalent; used for the cpu_baseline timing
 * and cross-checked against oracle/obs_oracle.py, which is pinned by the reference-generated golden vectors) */
static void euler_xyz_from_mat(const double* R, double* e) { /* scipy Rotation.as_euler('xyz'), extrinsic */
  double sy = -R[6];
  if (sy > 1) sy = 1; if (sy < -1) sy = -1;
  e[1] = asin(sy);
  if (fabs(sy) < 1 - 1e-12) { e[0] = atan2(R[7], R[8]); e[2] = atan2(R[3], R[0]); }
  else { e[0] = 0; e[2] = atan2(-R[1], R[4]); } /* gimbal lock: scipy sets the third angle to zero */
}

int gqo_get_obs(const GqOracle* o, const double* cmd /*[4]*/, const int* legs_order /*[4]*/, const int* obs_ids,
                int n_obs, double* out, int* terminated, int* invalid_contact) {
  const GqModelDesc* m = &o->d;
  double q[4], R[9], e[3], Rh[9];
  memcpy(q, o->qpos + 3, sizeof q);
  quat_normalize(q); /* scipy Rotation.from_quat normalises */
  quat2mat(R, q);
  euler_xyz_from_mat(R, e);
  double cy = cos(e[2]), sy = sin(e[2]);
  double Rh_[9] = {cy, -sy, 0, sy, cy, 0, 0, 0, 1};
  memcpy(Rh, Rh_, sizeof Rh);
  double tl[3], ta[3] = {0, 0, cmd[3]};
  mulmatvec3(tl, Rh, cmd);
  /* feet quantities from the (stale) position stage */
  double fpos[4][3], fvel[4][3], fvel_rel[4][3], cf[4][3];
  int cstate[4] = {0, 0, 0, 0};
  for (int l = 0; l < 4; l++) {
    int g = m->feet_geomid[l], b = m->geom_bodyid[g];
    memcpy(fpos[l], o->geom_xpos[g], sizeof fpos[l]);
    double jp[3][NV];
    gqo_jac(o, jp, NULL, fpos[l], b);
    for (int k = 0; k < 3; k++) {
      double s = 0;
      for (int i = 0; i < NV; i++) s += jp[k][i] * o->qvel[i];
      fvel[l][k] = s;
    }
    double d[3], cr[3];
    for (int k = 0; k < 3; k++) d[k] = fpos[l][k] - o->qpos[k];
    cross3(cr, o->qvel + 3, d); /* quirk B4: body-frame omega used as if world-frame (quadruped_env.py:659,669) */
    for (int k = 0; k < 3; k++) fvel_rel[l][k] = fvel[l][k] - o->qvel[k] - cr[k];
    cf[l][0] = cf[l][1] = cf[l][2] = 0;
  }
  *invalid_contact = 0;
  for (int c = 0; c < o->ncon; c++) {
    if (o->contact[c].body1 > 0) continue; /* contact between two bodies of the robot: "do nothing for now" (quadruped_env.py:1245-1246, :841) */
    int g = o->contact[c].geom, b = m->geom_bodyid[g], leg = -1;
    for (int l = 0; l < 4; l++)
      if (m->geom_bodyid[m->feet_geomid[l]] == b) leg = l;
    if (leg < 0) { *invalid_contact = 1; continue; }
    cstate[leg] = 1;
    double f6[6], fw[3];
    gqo_contact_force(o, c, f6);
    mulmatTvec3(fw, o->contact[c].frame, f6);
    for (int k = 0; k < 3; k++) cf[leg][k] += fw[k];
  }
  *terminated = *invalid_contact || o->qpos[0] > m->terrain_limits[0] || o->qpos[0] < m->terrain_limits[1] ||
                o->qpos[1] > m->terrain_limits[2] || o->qpos[1] < m->terrain_limits[3];
  double Mv[NV], Ma[NV], ke = 0, work = 0;
  for (int i = 0; i < NV; i++) {
    double s = 0, t = 0;
    for (int k = 0; k < NV; k++) { s += o->M[i][k] * o->qvel[k]; t += o->M[i][k] * o->qacc[k]; }
    Mv[i] = s; Ma[i] = t;
  }
  for (int i = 0; i < NV; i++) { ke += 0.5 * o->qvel[i] * Mv[i]; work += Ma[i] * o->qvel[i]; }
  double* p = out;
  double tmp[3], tmp2[3];
#define PUT3(v) { p[0] = (v)[0]; p[1] = (v)[1]; p[2] = (v)[2]; p += 3; }
  for (int n = 0; n < n_obs; n++) {
    switch (obs_ids[n]) {
      case GQ_OBS_BASE_POS: PUT3(o->qpos) break;
      case GQ_OBS_BASE_LIN_VEL: PUT3(o->qvel) break;
      case GQ_OBS_BASE_LIN_VEL_ERR: for (int k = 0; k < 3; k++) tmp[k] = tl[k] - o->qvel[k]; PUT3(tmp) break;
      case GQ_OBS_BASE_LIN_ACC: PUT3(o->qacc) break;
      case GQ_OBS_BASE_ANG_VEL: mulmatvec3(tmp, R, o->qvel + 3); PUT3(tmp) break;
      case GQ_OBS_BASE_ANG_VEL_ERR: mulmatvec3(tmp, R, o->qvel + 3); for (int k = 0; k < 3; k++) tmp[k] = ta[k] - tmp[k]; PUT3(tmp) break;
      case GQ_OBS_BASE_ORI_EULER_XYZ: PUT3(e) break;
      case GQ_OBS_BASE_ORI_QUAT_WXYZ: memcpy(p, o->qpos + 3, 4 * sizeof(double)); p += 4; break;
      case GQ_OBS_BASE_ORI_SO3: memcpy(p, R, sizeof R); p += 9; break;
      case GQ_OBS_GRAVITY_VECTOR_B: { double g[3] = {0, 0, -1}; mulmatTvec3(tmp, R, g); PUT3(tmp) } break;
      case GQ_OBS_BASE_LIN_VEL_B: mulmatTvec3(tmp, R, o->qvel); PUT3(tmp) break;
      case GQ_OBS_BASE_LIN_VEL_ERR_B: mulmatTvec3(tmp, R, tl); mulmatTvec3(tmp2, R, o->qvel); for (int k = 0; k < 3; k++) tmp[k] -= tmp2[k]; PUT3(tmp) break;
      case GQ_OBS_BASE_LIN_ACC_B: mulmatTvec3(tmp, R, o->qacc); PUT3(tmp) break;
      case GQ_OBS_BASE_ANG_VEL_B: PUT3(o->qvel + 3) break;
      case GQ_OBS_BASE_ANG_VEL_ERR_B: mulmatTvec3(tmp, R, ta); for (int k = 0; k < 3; k++) tmp[k] -= o->qvel[3 + k]; PUT3(tmp) break;
      case GQ_OBS_QPOS: memcpy(p, o->qpos, sizeof o->qpos); p += NQ; break;
      case GQ_OBS_QVEL: memcpy(p, o->qvel, sizeof o->qvel); p += NV; break;
      case GQ_OBS_TAU_CTRL_SETPOINT: memcpy(p, o->ctrl, sizeof(double) * m->nu); p += m->nu; break;
      case GQ_OBS_QPOS_JS: memcpy(p, o->qpos + 7, sizeof(double) * 12); p += 12; break;
      case GQ_OBS_QVEL_JS: memcpy(p, o->qvel + 6, sizeof(double) * 12); p += 12; break;
      case GQ_OBS_KINETIC_ENERGY: *p++ = ke; break;
      case GQ_OBS_WORK: *p++ = work; break;
      case GQ_OBS_FEET_POS: for (int l = 0; l < 4; l++) PUT3(fpos[legs_order[l]]) break;
      case GQ_OBS_FEET_POS_B: for (int l = 0; l < 4; l++) { for (int k = 0; k < 3; k++) tmp2[k] = fpos[legs_order[l]][k] - o->qpos[k]; mulmatTvec3(tmp, R, tmp2); PUT3(tmp) } break;
      case GQ_OBS_FEET_VEL: for (int l = 0; l < 4; l++) PUT3(fvel[legs_order[l]]) break;
      case GQ_OBS_FEET_VEL_REL: for (int l = 0; l < 4; l++) PUT3(fvel_rel[legs_order[l]]) break;
      case GQ_OBS_FEET_VEL_B: for (int l = 0; l < 4; l++) { mulmatTvec3(tmp, R, fvel[legs_order[l]]); PUT3(tmp) } break;
      case GQ_OBS_FEET_VEL_REL_B: for (int l = 0; l < 4; l++) { mulmatTvec3(tmp, R, fvel_rel[legs_order[l]]); PUT3(tmp) } break;
      case GQ_OBS_CONTACT_STATE: for (int l = 0; l < 4; l++) *p++ = cstate[l]; break; /* quirk B5: always FL FR RL RR */
      case GQ_OBS_CONTACT_FORCES: for (int l = 0; l < 4; l++) PUT3(cf[legs_order[l]]) break;
      case GQ_OBS_CONTACT_FORCES_B: for (int l = 0; l < 4; l++) { mulmatTvec3(tmp, R, cf[legs_order[l]]); PUT3(tmp) } break;
      default: snprintf(g_err, sizeof g_err, "gqo_get_obs: bad obs id %d", obs_ids[n]); return GQ_EINVAL;
    }
  }
  return (int)(p - out);
}

/* cpu_baseline helper: run `nsteps` steps with the given control sequence [nsteps][nu], assembling the observation
 * row every step as QuadrupedEnv.step does; when reset_on_term != 0 a terminated env is put back to the state the
 * rollout started from (stand-in for the user's env.reset()).  Returns the number of terminated steps. */
int gqo_rollout(GqOracle* o, const double* ctrl_seq, int nsteps, const double* cmd, const int* legs_order,
                const int* obs_ids, int n_obs, double* obs_last, int reset_on_term) {
  static double buf[512];
  double q0[NQ], v0[NV];
  memcpy(q0, o->qpos, sizeof q0); memcpy(v0, o->qvel, sizeof v0);
  int term, inv, nterm = 0;
  for (int s = 0; s < nsteps; s++) {
    gqo_step(o, ctrl_seq + (size_t)s * o->d.nu);
    gqo_get_obs(o, cmd, legs_order, obs_ids, n_obs, obs_last ? obs_last : buf, &term, &inv);
    nterm += term;
    if (term && reset_on_term) {
      memcpy(o->qpos, q0, sizeof q0); memcpy(o->qvel, v0, sizeof v0);
      memset(o->qacc_warmstart, 0, sizeof o->qacc_warmstart);
    }
  }
  return nterm;
}
