/*
 * gq.h - C-ABI of libgq, the MI355X-native batched replacement for the MuJoCo
 * calls on gym-quadruped's QuadrupedEnv.step() hot path.
 *
 * Every entry point names the reference interface it replaces.  Reference
 * paths are relative to the gym-quadruped repository (iit-DLSLab/gym-quadruped
 * v1.1.5); "mujoco.*" is the third-party pybind API the reference calls.
 *
 *   reference call site                                   replaced by
 *   ----------------------------------------------------  --------------------
 *   mujoco.MjModel.from_xml_path   quadruped_env.py:170    gq_model_create
 *   mujoco.MjData(model)           quadruped_env.py:178    gq_batch_create
 *   mujoco.mj_step                 quadruped_env.py:271    gq_step  (+ the
 *     _get_obs / _check_* epilogue quadruped_env.py:277-285, :1146-1257)
 *   mujoco.mj_step1                quadruped_env.py:376    gq_reset (lift loop)
 *   mj_resetDataKeyframe + reset   quadruped_env.py:343-397 gq_reset
 *   mujoco.mj_jac                  quadruped_env.py:728    gq_jac (+ gq_step obs epilogue for the feet)
 *   mujoco.mj_step1 / mj_forward   quadruped_env.py:376,384,1321  gq_forward
 *   mujoco.mj_ray                  sensors/heightmap.py:90-99     gq_ray (general rays), gq_heightmap (the HeightMap grid)
 *   mujoco.mj_contactForce         quadruped_env.py:852    gq_contact_force (contact rows of gq_batch_set_outputs); summed per foot: obs epilogue
 *   mujoco.mj_fullM                quadruped_env.py:940    dyn rows of gq_batch_set_outputs (production kernel); gq_full_mass (inspection record)
 *
 * Conventions
 *  - plain C, no exceptions cross the boundary; every function returns 0 on
 *    success or a negative GQ_E* code, text via gq_last_error().
 *  - all batch tensors are CALLER-owned device memory (PyTorch-ROCm
 *    allocations), passed as raw pointers; the library owns only the opaque
 *    GqModel / GqBatch handles (model constants + debug scratch on device).
 *  - batch layout: one array per field, env-major rows:  field[env][dim].
 *    One wavefront owns one env, so each row is read/written by consecutive
 *    lanes = one coalesced <=128 B segment per access (DESIGN.md "HBM layout").
 *  - every launch is asynchronous on the hipStream_t passed in (as void*);
 *    no hidden synchronisation.  Handles are not thread-safe.
 */
#ifndef GQ_H_
#define GQ_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GQ_OK 0
#define GQ_EINVAL (-1)   /* bad argument / unsupported model feature */
#define GQ_ENOMEM (-2)
#define GQ_EDEVICE (-3)  /* HIP runtime error */
#define GQ_ENODEVICE (-4)

#define GQ_NLEG 4
#define GQ_MAX_NV 18
#define GQ_MAX_NBODY 14

/* ---- model description: flat host arrays produced by the MJCF compiler -------------
 * (gym_quadruped_amd/mjcf.py; field names follow mujoco.MjModel).  All pointers are
 * host memory, read during gq_model_create only. */
/* ABI revision of this header.  gq_version() returns the revision the LIBRARY was built from: a binding compares the two
 * before its first call (gym_quadruped_amd/_lib.py does), and gq_struct_sizes() lets it check its own mirror of every struct
 * that crosses the boundary.  History: 100 round 1; 300 = GqModelDesc.struct_size + the self-collision / geom_type tables,
 * GqObsOut.step_num_prev, strided HeightMap views, the round-3 entry points; 400 = the closed-loop persistent rollout
 * (gq_rollout_closed, gq_mailbox_get), GqObsOut.contacts_dropped; 500 = lap-tagged mailbox queue items, gq_struct_sizes(out[8]),
 * GqModelDesc.plane_* (optional); 510 = gq_batch_set_heightmap (no struct changed); 600 = GqModelDesc.vert_adj* / plane_order (hull
 * graphs: multi-point mesh-plane contacts), the general convex narrow phase (GJK / EPA) behind the same tables; 610 = gq_batch_set_pair_exchange
 * (no struct changed); 620 = GqModelDesc.support_grid (optional); 630 = that grid 16 x 16 cells per face (was 8 x 8). */
#define GQ_ABI_VERSION 630
#ifndef GQ_SUPPORT_GRID
#define GQ_SUPPORT_GRID 16 /* cells per edge of a cube-map face of GqModelDesc.support_grid */
#endif

typedef struct GqModelDesc {
  int32_t struct_size; /* = sizeof(GqModelDesc) of the caller's header; gq_model_create refuses any other value */
  /* sizes */
  int32_t nq, nv, nu, nbody, njnt, ngeom, ncloud, nvert;
  /* options (mjOption) */
  double timestep;
  double gravity[3];
  int32_t cone;        /* 0 pyramidal, 1 elliptic */
  double impratio;
  int32_t integrator;  /* 0 Euler (3 = implicitfast is run as Euler + implicit joint damping) */
  /* bodies */
  const int32_t* body_parentid; /* [nbody] */
  const double* body_pos;       /* [nbody][3] */
  const double* body_quat;      /* [nbody][4] wxyz */
  const double* body_ipos;      /* [nbody][3] */
  const double* body_iquat;     /* [nbody][4] */
  const double* body_mass;      /* [nbody] */
  const double* body_inertia;   /* [nbody][3] */
  const int32_t* body_jntadr;   /* [nbody] */
  const int32_t* body_jntnum;   /* [nbody] */
  const double* body_invweight0;/* [nbody][2] */
  /* joints */
  const int32_t* jnt_type;      /* [njnt] 0 free, 3 hinge */
  const int32_t* jnt_bodyid;
  const int32_t* jnt_qposadr;
  const int32_t* jnt_dofadr;
  const double* jnt_pos;        /* [njnt][3] */
  const double* jnt_axis;       /* [njnt][3] */
  const int32_t* jnt_limited;
  const double* jnt_range;      /* [njnt][2] */
  const double* jnt_margin;
  const double* jnt_solref;     /* [njnt][2] limit */
  const double* jnt_solimp;     /* [njnt][5] limit */
  const int32_t* jnt_actfrclimited;
  const double* jnt_actfrcrange;/* [njnt][2] */
  const double* qpos0;          /* [nq] AFTER the registry's qpos0_js override (robot_cfgs.py:39, quadruped_env.py:171-173) */
  /* dofs */
  const int32_t* dof_bodyid;    /* [nv] */
  const int32_t* dof_jntid;
  const int32_t* dof_parentid;
  const double* dof_damping;
  const double* dof_armature;
  const double* dof_frictionloss;
  const double* dof_solref;     /* [nv][2] friction loss */
  const double* dof_solimp;     /* [nv][5] */
  const double* dof_invweight0;
  /* robot collision geoms lowered to vertex clouds (geoms of body 0 are excluded) */
  const int32_t* geom_bodyid;   /* [ngeom] */
  const double* geom_pos;       /* [ngeom][3] body frame */
  const double* geom_quat;      /* [ngeom][4] */
  const int32_t* geom_cloudid;  /* [ngeom] -1 = does not collide */
  const double* geom_friction;  /* [ngeom][3] */
  const double* geom_margin;
  const double* geom_gap;
  const int32_t* geom_condim;
  const int32_t* geom_priority;
  const double* geom_solref;    /* [ngeom][2] */
  const double* geom_solimp;    /* [ngeom][5] */
  const double* geom_solmix;
  const double* geom_rbound;
  const int32_t* cloud_vertadr; /* [ncloud] */
  const int32_t* cloud_vertnum;
  const double* cloud_radius;
  const double* vert_pos;       /* [nvert][3] geom frame */
  /* actuators (torque motors) */
  const int32_t* actuator_trnid;   /* [nu] joint id */
  const double* actuator_gear;
  const int32_t* actuator_ctrllimited;
  const double* actuator_ctrlrange;/* [nu][2] */
  const int32_t* actuator_forcelimited;
  const double* actuator_forcerange;
  /* scene: ground plane z = 0 named "floor" (utils/mujoco/assets/scene_flat.xml:31) */
  double floor_friction[3];
  double floor_margin, floor_gap, floor_solmix;
  double floor_solref[2];
  double floor_solimp[5];
  int32_t floor_condim, floor_priority;
  /* env-level constants */
  int32_t feet_geomid[GQ_NLEG];  /* FL FR RL RR (robot_cfgs.py:15) */
  double terrain_limits[4];      /* max_x min_x max_y min_y (terrain.py:359) */
  double meaninertia;
  double key_qpos[19];           /* keyframe 0 (mj_resetDataKeyframe, quadruped_env.py:343) */
  /* solver */
  int32_t solver;                /* 0 PGS (mj_solPGS, named by the north-star; pyramidal cones, every scene), 1 Newton (mj_solNewton, MuJoCo's default) */
  int32_t iterations;
  double tolerance;
  /* fp32 stopping rule of the Newton solver (no MuJoCo counterpart; 0 = off): after at least one Newton step the
   * iteration also stops when |grad| <= noise_floor * sqrt(|M dq|^2 + |J'f|^2), i.e. when the gradient - a
   * difference of those two vectors - is down at the round-off of its own terms and `tolerance` (meant for fp64
   * magnitudes) can no longer be met by anything but a wasted extra iteration. */
  double noise_floor;
  /* static world boxes of the scene (terrain.py add_box :121-142: random_boxes, random_pyramids, ramp, slippery, stairs):
   * world geoms 1..nbox after the floor, each with MuJoCo's per-geom contact parameters.  box_mat is the row-major
   * rotation (columns = box axes in the world), box_size the half extents. */
  int32_t nbox;
  const double* box_pos;        /* [nbox][3] */
  const double* box_mat;        /* [nbox][9] */
  const double* box_size;       /* [nbox][3] */
  const double* box_friction;   /* [nbox][3] */
  const double* box_margin;     /* [nbox] */
  const double* box_gap;
  const double* box_solmix;
  const double* box_solref;     /* [nbox][2] */
  const double* box_solimp;     /* [nbox][5] */
  const int32_t* box_condim;
  const int32_t* box_priority;
  /* height field of the scene (terrain.py add_perlin_heightfield :26-113; MuJoCo hfield geom, identity orientation):
   * hfield_data holds the elevation normalised to [0, 1] as MuJoCo's compiler leaves it, [nrow][ncol] with row r at
   * y = -size[1] + r * 2 size[1] / (nrow - 1) and column c likewise along x; world elevation = pos[2] + data * size[2].
   * hfield_nrow = 0: no height field.  Contact parameters as for any geom.  Needs the Newton solver. */
  int32_t hfield_nrow, hfield_ncol;
  const float* hfield_data;
  double hfield_size[4];        /* radius_x, radius_y, elevation_z, base_z */
  double hfield_pos[3];
  double hfield_friction[3], hfield_margin, hfield_gap, hfield_solmix, hfield_solref[2], hfield_solimp[5];
  int32_t hfield_condim, hfield_priority;
  /* robot self-collision (mj_collision between two bodies of the robot; the reference models keep MuJoCo's default
   * contype = conaffinity = 1, e.g. aliengo.xml:8-10,42,61,71, mini_cheetah.xml:33-35,66,75,92, spot.xml:177-186 excludes):
   * the geom pairs that pass MuJoCo's static filter (different bodies, contype / conaffinity, not parent and child, not
   * excluded), ordered by (body1, body2, geom1, geom2), and one proxy capsule per collision geom in its BODY frame
   * (p0[3], p1[3], radius): exact for sphere / capsule geoms, the INSCRIBED capsule along the hull's principal axis for box /
   * cylinder / mesh geoms (a contact is found late by the gap between hull and proxy, never invented;
   * gym_quadruped_amd/selfcol.py).  The proxies decide pairs of hull / cylinder geoms; a pair of a BOX with a sphere, capsule
   * or box (geom_type below) is evaluated exactly instead - closest features, separating axes, up to 2 / 4 points
   * (csrc/gq_pairs.h).  nselfpair = 0 switches self-collision off (it is ON by default with the Newton solver). */
  int32_t nselfpair;
  const int32_t* selfpair_geom1; /* [nselfpair] */
  const int32_t* selfpair_geom2;
  const double* geom_capsule;    /* [ngeom][7] */
  /* MuJoCo geom type (mjtGeom: 2 sphere, 3 capsule, 5 cylinder, 6 box, 7 mesh) of every geom: selects the routine of the
   * plane narrow phase, in the kernel (csrc/gq_step_body.h floor_candidates) and in the oracle alike - mjraw_PlaneCapsule:
   * both end spheres, frame aligned with the axis; mjraw_PlaneBox: the corners at or below the centre, at most 4;
   * mjc_PlaneCylinder: up to 4 rim points; mesh: the support vertex of the hull (mjc_PlaneConvex's first point).
   * It also selects the exact pair routines for sphere / capsule / box geoms against world boxes and against each other.
   * NULL: spheres and capsules are recognised by their clouds (1 / 2 vertices), everything else is a hull. */
  const int32_t* geom_type;      /* [ngeom] */
  /* OPTIONAL acceleration structure of the hull-versus-plane narrow phase (mjc_PlaneConvex's support vertex: the hull vertex deepest
   * along the plane normal); plane_grid = 0 / NULL pointers: every 64-vertex chunk of a cloud is scanned.
   * plane_vert_pos: the clouds' vertices once more, every cloud (same cloud_vertadr / cloud_vertnum) in DIRECTION order, so that a chunk
   * of 64 consecutive vertices answers a patch of directions.  plane_mask: per cloud and per cell of a cube map of the unit sphere
   * (cell = ((axis * 2 + negative) * G + iu) * G + iv: dominant axis of the direction and its sign, then the other two coordinates
   * divided by the dominant one, rastered G x G over [-1, 1]^2), bit k is set when chunk k may hold the support vertex of SOME
   * direction in the cell - a superset is fine, a missing bit is a missed contact.  gym_quadruped_amd/cabi.py plane_support_tables
   * builds both from the hulls' facet normals. */
  int32_t plane_grid;            /* G (<= 16), 0: no tables */
  const double* plane_vert_pos;  /* [nvert][3] */
  const int32_t* plane_mask;     /* [ncloud][6 G G] */
  /* OPTIONAL hull graphs of the mesh clouds (MuJoCo's mesh_graph): the edges of the convex hull, per vertex the list of the vertices it
   * shares an edge with (cloud-LOCAL indices, ascending).  mjc_PlaneConvex walks them: after the support vertex of a mesh against a
   * plane, the neighbours within the margin become contacts too, until the pair has three (csrc/gq_step_body.h stage_collision_scan,
   * oracle/gq_oracle.c gqo_collision).  vert_adjnum = 0 for the vertices of clouds that are not meshes; NULL pointers: support vertex
   * only.  plane_order (with plane_vert_pos): cloud-local index, in vert_pos, of every vertex of the direction-ordered copy.
   * gym_quadruped_amd/cabi.py hull_graphs builds them with scipy.spatial.ConvexHull. */
  int32_t nadj;
  const int32_t* vert_adjadr;    /* [nvert] first entry of the vertex's list in vert_adj */
  const int32_t* vert_adjnum;    /* [nvert] */
  const int32_t* vert_adj;       /* [nadj] */
  const int32_t* plane_order;    /* [nvert] */
  /* OPTIONAL (with plane_vert_pos): per cloud and per 64-vertex chunk of its direction-ordered copy a cap of directions - unit axis (geom
   * frame), cosine of the half angle - containing every direction one of the chunk's vertices supports; cos = -2: the chunk is always
   * scanned.  The convex routine (csrc/gq_convex.h) takes its support vertices from the chunks whose cap contains the query direction. */
  const double* plane_cap;       /* [ncloud][16][4] */
  /* OPTIONAL: per cloud the support function h(u) = max over its vertices v of v . u (geom frame) at the nodes of a cube map - face f of
   * +x, -x, +y, -y, +z, -z, node (i, j), i, j = 0..GQ_SUPPORT_GRID: u = the face's axis with the two other coordinates (in x, y, z order) at -1 + 2 i / GQ_SUPPORT_GRID
   * and -1 + 2 j / GQ_SUPPORT_GRID, NOT normalised.  h is convex and positively homogeneous, so the bilinear blend of a cell's four nodes is an UPPER bound
   * of h inside the cell (second order in the cell size: millimetres on a robot link): a lane tells a hull pair apart - no contact within
   * the margin - without a support query of the convex routine (csrc/gq_convex.h cvx_hgrid).  NULL: every pair past the oriented boxes
   * goes to the routine.  Values must not be below the true maxima (gym_quadruped_amd/cabi.py support_grids rounds up). */
  const double* support_grid;    /* [ncloud][6][GQ_SUPPORT_GRID + 1][GQ_SUPPORT_GRID + 1] */
  /* Robot-robot pairs that involve a mesh or a cylinder: 1 = the general convex routine on the geoms' hulls (mjc_Convex: GJK distance, EPA
   * penetration - what MuJoCo computes; csrc/gq_convex.h), 0 = the capsules of geom_capsule in their place (an approximation that finds such a
   * contact late, by the gap between hull and capsule, and costs a closest-point computation of two segments instead of ten to twenty support
   * queries of two hulls - the flat-scene step runs three to ten times faster on a batch whose robots tangle their legs).  World boxes and the
   * floor are not affected: meshes meet them through the convex routine / the hull graph either way. */
  int32_t self_convex;
} GqModelDesc;

typedef struct GqModel GqModel;
typedef struct GqBatch GqBatch;

/* ---- per-env state, device pointers, env-major rows --------------------------------- */
typedef struct GqState {
  double* qpos;            /* [N][19] f64: base xyz must survive |xy| ~ 1e4 m (terrain.py:359)      */
  float* qvel;             /* [N][18]                                                               */
  float* qacc;             /* [N][18] out: mjData.qacc of the last forward pass                     */
  float* qacc_warmstart;   /* [N][18] in/out                                                        */
  float* qfrc_applied;     /* [N][18] in: external disturbance wrench (quadruped_env.py:305)        */
  float* time;             /* [N] in/out                                                            */
  float* friction;         /* [N] tangential coeff. of floor + feet (quadruped_env.py:1277-1298); < 0 = XML values */
  float* cmd;              /* [N][4] ref_base_lin_vel_H[3], ref_base_ang_yaw_dot (:1046-1072)       */
} GqState;

/* ---- observation outputs: one contiguous row per env -------------------------------- */
typedef struct GqObsOut {
  float* obs;              /* [N][obs_dim] concatenation in the order of obs_ids                    */
  float* reward;           /* [N]   (_compute_reward: constant 0, quadruped_env.py:1141)            */
  uint8_t* terminated;     /* [N]   invalid contact | out of terrain bounds (:283-285)              */
  uint8_t* truncated;      /* [N]   always 0 (:286)                                                 */
  uint8_t* invalid_contact;/* [N]   info['invalid_contacts'] non-empty (:1228-1248)                 */
  int32_t* step_num;       /* [N]   in/out                                                          */
  int32_t* step_num_prev;  /* [N]   out, may be NULL: the counter BEFORE this step's increment = info['step_num'] (:288-290) */
  int32_t* contacts_dropped;/* [N]  out, may be NULL: contacts the narrow phase found this step that did NOT enter the constraint set
                            * because the env was at the kernel's capacity (12 contacts / 63 rows; MuJoCo's mj_step, :271, has no
                            * such cap) - 0 means the step saw every contact                                                   */
} GqObsOut;

/* observable ids = index into QuadrupedEnv.ALL_OBS (quadruped_env.py:35-66,81) */
enum GqObsId {
  GQ_OBS_BASE_POS = 0, GQ_OBS_BASE_LIN_VEL, GQ_OBS_BASE_LIN_VEL_ERR, GQ_OBS_BASE_LIN_ACC, GQ_OBS_BASE_ANG_VEL,
  GQ_OBS_BASE_ANG_VEL_ERR, GQ_OBS_BASE_ORI_EULER_XYZ, GQ_OBS_BASE_ORI_QUAT_WXYZ, GQ_OBS_BASE_ORI_SO3,
  GQ_OBS_GRAVITY_VECTOR_B,
  GQ_OBS_BASE_LIN_VEL_B, GQ_OBS_BASE_LIN_VEL_ERR_B, GQ_OBS_BASE_LIN_ACC_B, GQ_OBS_BASE_ANG_VEL_B,
  GQ_OBS_BASE_ANG_VEL_ERR_B,
  GQ_OBS_QPOS, GQ_OBS_QVEL, GQ_OBS_TAU_CTRL_SETPOINT, GQ_OBS_QPOS_JS, GQ_OBS_QVEL_JS, GQ_OBS_KINETIC_ENERGY,
  GQ_OBS_WORK,
  GQ_OBS_FEET_POS, GQ_OBS_FEET_POS_B, GQ_OBS_FEET_VEL, GQ_OBS_FEET_VEL_REL, GQ_OBS_FEET_VEL_B,
  GQ_OBS_FEET_VEL_REL_B, GQ_OBS_CONTACT_STATE, GQ_OBS_CONTACT_FORCES, GQ_OBS_CONTACT_FORCES_B,
  /* sensor observables (sensors/imu.py:17-18, LIN_ACC_OBS + GYRO_OBS); valid after gq_batch_set_imu */
  GQ_OBS_IMU_ACC, GQ_OBS_IMU_ACC_NOISE, GQ_OBS_IMU_ACC_BIAS, GQ_OBS_IMU_GYRO, GQ_OBS_IMU_GYRO_NOISE, GQ_OBS_IMU_GYRO_BIAS,
  GQ_OBS_COUNT
};

const char* gq_last_error(void);
int gq_version(void);   /* GQ_ABI_VERSION of the library build */
/* sizeof of the structs that cross this boundary, in the library's build: out[0..7] = GqModelDesc, GqState, GqObsOut,
 * GqResetCfg, GqResampleCfg, GqImuCfg, GqPolicyPd, GqMailboxView.  A binding whose mirror of one of them differs must not call the library. */
int gq_struct_sizes(int32_t out[8]);

/* dimension of observable `id` (19,18,12,... as configure_observation_space, quadruped_utils.py:235-325) */
int gq_obs_dim(int obs_id);

/* mujoco.MjModel.from_xml_path (quadruped_env.py:170): uploads model constants to `device` */
int gq_model_create(const GqModelDesc* desc, int device, GqModel** out);
int gq_model_destroy(GqModel* m);

/* mujoco.MjData (quadruped_env.py:178): per-batch launch geometry + debug scratch.
 * obs_ids/n_obs = state_obs_names; legs_order = permutation of {0:FL,1:FR,2:RL,3:RR} (quadruped_env.py:95) */
int gq_batch_create(GqModel* m, int n_envs, const int32_t* obs_ids, int n_obs, const int32_t* legs_order,
                    GqBatch** out);
int gq_batch_destroy(GqBatch* b);
int gq_batch_obs_dim(const GqBatch* b);

/* IMU sensor plug-in (sensors/imu.py:20-139): accelerometer + gyro at a site of the base body, white noise and
 * random-walk bias.  Replaces IMU.step()'s reads of mjData.sensordata (mj_sensorAcc / mj_sensorVel outputs) and its
 * np.random.normal draws (counter-based Philox normals here; draw order acc noise, acc bias step, gyro noise, gyro
 * bias step as in compute_linear_acceleration / compute_angular_velocity).  bias_state: device [N][6] in/out
 * (accelerometer bias xyz, gyro bias xyz), caller-owned, persists across resets like the reference's IMU object. */
typedef struct GqImuCfg {
  double site_pos[3];    /* site in the base-body frame (<site name="imu" .../>) */
  double site_quat[4];
  float accel_noise, gyro_noise, accel_bias_rate, gyro_bias_rate;
  uint64_t seed;
} GqImuCfg;
int gq_batch_set_imu(GqBatch* b, const GqImuCfg* cfg, float* bias_state);

/* HeightMap that follows the base (sensors/heightmap.py:17-221 as examples/aliengo_with_heightmap.py uses it: update_height_map(qpos[0:3],
 * yaw = base_ori_euler_xyz[2]) after every step).  From the next gq_step on, the step kernel itself casts the rows x cols rays of every env it
 * steps - grid centred on the NEW base position, heading = yaw of the new base orientation (scipy as_euler('xyz')[2], as the observation row
 * holds it) - and writes the hit points to out, device [N][rows * cols][3] (caller-owned; cell order and ray geometry as gq_heightmap): no
 * kernel and no launch boundary of its own behind the step.  An env the step's mask leaves out keeps its rows.  Scenes with world boxes or a
 * height field only (GQ_EINVAL on a flat scene: gq_heightmap there).  out = NULL switches it off.  The heading is taken from the base
 * rotation's (R10, R00), normalised - the yaw of as_euler('xyz') away from the gimbal pole.  Every forward pass of the step kernel that
 * advances an env casts its rays, the reset's own mj_step (gq_reset, the in-kernel auto-reset) included; what gq_reset's step wrote is the
 * map of the state AFTER that step, so a caller that resets with an explicit state, restores a snapshot or writes the state tensors in
 * place launches gq_heightmap once for the state it now holds (the Python HeightMap does: it ties the map's freshness to the state tensor's
 * version).  No reference counterpart as a call: the reference casts mj_ray per cell from Python. */
int gq_batch_set_heightmap(GqBatch* b, int rows, int cols, float dist_x, float dist_y, float* out);

/* The convex pair exchange of a batch (gym_quadruped_amd/csrc/gq_exchange.h): in a full-batch launch (gq_step, the persistent gq_rollout) an env
 * with several hull pairs in reach hands all but one of them to wavefronts of the same launch that have finished their own env - the launch
 * no longer lasts as long as its most entangled robot.  Results are bit-identical either way (a pair's contact does not depend on which
 * wavefront computes it).  On by default for a model with convex self pairs (GqModelDesc.self_convex); on = 0 switches it off (A/B
 * measurements, tests), on = 1 back on.  Returns GQ_OK, or GQ_EINVAL for a model without such pairs when on = 1.  No reference counterpart
 * (MuJoCo's narrow phase is serial per mjData). */
int gq_batch_set_pair_exchange(GqBatch* b, int on);

/* reset configuration: the knobs of QuadrupedEnv.reset / _sample_ref_vel / _set_ground_friction */
typedef struct GqResetCfg {
  uint64_t seed;            /* key of the counter-based device RNG (Philox4x32-10; counter = draw, episode, env) */
  int32_t random;           /* reset(random=...) (quadruped_env.py:314,346) */
  float q_pos_amp;          /* joint angle noise amplitude, default 20 deg (:347) */
  float q_vel_amp;          /* joint velocity noise amplitude 0.5 (:348) */
  float roll_sweep, pitch_sweep; /* default 10 deg (:362-363) */
  float hip_height;         /* RobotConfig.hip_height: spawn height before the lift loop (:359) */
  float lin_vel_range[2];   /* ref_base_lin_vel (min,max) (:141) */
  float ang_vel_range[2];   /* ref_base_ang_vel (min,max) (:142) */
  float friction_range[2];  /* ground_friction_coeff (min,max) (:143) */
  int32_t cmd_forward, cmd_random, cmd_rotate, cmd_human; /* substrings of base_vel_command_type (:1049-1066) */
  int32_t env_id_offset;    /* global id of env 0 of this batch (multi-GPU shards draw from disjoint counters) */
  int32_t autoreset_next_step; /* only read by gq_step(auto_reset=...): 0 = same-step auto-reset, 1 = next-step (see gq_step) */
} GqResetCfg;

/* In-episode resampling of the velocity command and of the external disturbance wrench (QuadrupedEnv.step
 * quadruped_env.py:292-305, _sample_ref_vel :1046-1072, _sample_external_disturbances :1074-1139), folded into the
 * epilogue of gq_step for every env: `after += 1; if (after >= before) redraw` with `before ~ U{1000..2999}`.
 * counters: device [N][6] int32, caller-owned, in/out: {after_vel, before_vel, n_vel, after_dist, before_dist, n_dist}
 * (n_* = number of redraws so far: the RNG counter word).  ext_dist: device [N][6] f32 in/out, the current wrench
 * (x y z roll pitch yaw); when dist_reset != 0 every user step ends with qfrc_applied[:6] = ext_dist, which therefore acts
 * from the NEXT step on (:305).  The command redraw uses the knobs of the GqResetCfg passed here (ranges, cmd_* flags);
 * gq_reset / the in-kernel auto-reset restart the command interval of the envs they reset (:1068-1070).
 * Draws: Philox4x32-10, key = seed, counter = (draw / 4, n_*, global env id, 0xc0de) - see tests/philox_ref.py.
 * dist_kind[k]: 0 absent (0.0), 1 constant dist_range[k][0], 2 uniform in dist_range[k].  cfg NULL switches it off.
 * The change takes effect with the NEXT launch of this batch and is ordered on that launch's stream (no device access here). */
typedef struct GqResampleCfg {
  uint64_t seed;
  int32_t cmd_reset;        /* 'reset' in base_vel_command_type (:293) */
  int32_t dist_reset;       /* external_disturbances_kwargs['type'] == 'reset' (:299) */
  int32_t dist_kind[6];
  float dist_range[6][2];
  int32_t env_id_offset;
} GqResampleCfg;
int gq_batch_set_resampling(GqBatch* b, const GqResampleCfg* cfg, const GqResetCfg* cmd_cfg, int32_t* counters, float* ext_dist);

/* QuadrupedEnv.step body (quadruped_env.py:270-290): ctrl <- action; mj_step; _get_obs; reward; termination.
 * ctrl: device [N][nu] f32, or NULL for zero control.  mask: device [N] u8 or NULL - envs with mask==0 are left untouched
 * (used by reset(), which ends with one mj_step for the envs being reset, quadruped_env.py:397).
 * auto_reset != NULL: an env whose step terminates is re-spawned INSIDE the same launch (the batched stand-in for the
 * user's `if terminated: env.reset()` loop): reset state write + lift loop + the reset's own mj_step, exactly as
 * gq_reset does; its `terminated` flag stays set and its observation row is the first one of the new episode.
 * auto_reset->autoreset_next_step != 0 selects gymnasium's NEXT_STEP convention instead: the terminating step returns
 * the terminal observation, and the env spends its NEXT gq_step call on the reset (control ignored; state write + lift
 * loop + the reset's own mj_step; flags 0, observation = first of the new episode).  Every launch then runs exactly
 * one mj_step per env, so no wavefront does two passes.  The pending-reset flags live in the batch and are cleared by
 * gq_reset for the envs it resets.
 * episode / lift_failed as in gq_reset (may be NULL when auto_reset is NULL). */
int gq_step(GqBatch* b, const float* ctrl, const uint8_t* mask, GqState st, GqObsOut out,
            const GqResetCfg* auto_reset, int32_t* episode, uint8_t* lift_failed, void* hip_stream);

/* gq_step for the envs [env0, env0 + count) only (grid = count wavefronts); every tensor is still the full batch's.  This is
 * what an open-loop rollout is pipelined with: the batch is cut into a few shards, each shard's steps are chained on a HIP
 * stream of its own, and - envs being independent - no launch ever waits for another shard's stragglers
 * (QuadrupedEnv.rollout).  The argument block must already be bound to these tensors: call gq_batch_bind (or any gq_step)
 * on a stream the shard streams are ordered after. */
int gq_step_range(GqBatch* b, int env0, int count, const float* ctrl, GqState st, GqObsOut out, const GqResetCfg* auto_reset,
                  int32_t* episode, uint8_t* lift_failed, void* hip_stream);
/* K steps of every env with the action sequence ctrl_seq (device [K][N][nu] f32), equivalent to K gq_step calls but
 * pipelined over `shards` groups of envs on library-owned HIP streams (forked from / joined to hip_stream with events): no
 * launch waits for another group's stragglers or for the gap between two dependent launches.  obs_seq: device
 * [K][N][obs_dim] f32 receiving every step's observation rows, or NULL (the row tensor then holds the last step's).
 * shards = 0 selects the PERSISTENT form: one launch in which every wavefront plays all K steps of its env back to back - no
 * launch boundaries at all, so no env ever waits for the slowest env of a step (needs next-step auto-reset or none). */
int gq_rollout(GqBatch* b, const float* ctrl_seq, int n_steps, int shards, GqState st, GqObsOut out, const GqResetCfg* auto_reset,
               int32_t* episode, uint8_t* lift_failed, float* obs_seq, void* hip_stream);
/* CLOSED-LOOP persistent rollout: n_steps steps of every env with a policy in the loop and no launch boundary - the form of
 * `for k: action = policy(obs); obs, ... = env.step(action)` (quadruped_env.py:251-307 called from a control loop, README.md:31-33)
 * in which env e's step k + 1 waits for env e's action only, never for the slowest env of step k.  Env-steps are tasks:
 *   policy side   for each env e, for k = 0 .. n_steps - 1:  wait until steps_done[e] >= k (the observation row of e after k steps
 *                 is in GqObsOut.obs, written through to device-coherent memory); write the 12 torques to action[e][0..12);
 *                 then push e onto ready queue  e % n_queues:  s = atomic_add(counters[(3 q + 1) * counter_stride], 1);
 *                 items[q * queue_capacity + (s & (queue_capacity - 1))] = (((s / queue_capacity) & 127) << 24) | (e + 1)
 *                 (device-scope stores, the item last; the lap number of the push ticket rides in bits 24..30, so that pop tickets that
 *                 run more than queue_capacity ahead of the pushes - many resident step wavefronts, few envs - each take the item of
 *                 their own lap; N <= 2^24 - 1 envs, N / n_queues * n_steps < 2^31 tickets per queue);
 *   step side     the wavefronts of ONE launch pop tickets from the queue of their XCD, play one QuadrupedEnv.step() of the popped
 *                 env each (mj_step, observation / termination epilogue, next-step auto-reset exactly as gq_step), publish the
 *                 observation row and steps_done[e] = k + 1, and pop again until all N * n_steps env-steps are claimed.
 * Any number of envs works with any number of resident wavefronts (an env is not bound to a wavefront), and every wait has a
 * deadline: when it passes, status[0] becomes non-zero, every participant leaves, and gq_rollout_closed_status reports it - a
 * missing or stuck policy is an error, not a hang.  The final state and flags are those of n_steps gq_step calls with the same
 * actions, bit for bit.
 * pd != NULL: the library runs its built-in policy kernel (joint-space PD: torque_j = kp_j (q_des_j - q_j) - kd_j qd_j on the
 * qpos_js / qvel_js - or qpos / qvel - columns of the observation row, every operation rounded like the elementwise expression)
 * on a stream of its own, `policy_waves` wavefronts (0: default), launched BEFORE the step kernel and awaited until resident.
 * pd == NULL: the caller provides the policy side (gq_mailbox_get gives the device pointers) and must have it running.  XCD rule: the
 * wavefront that writes env e's action must run on the XCD of queue e % n_queues (GqMailboxView.xcc_queue maps HW_REG_XCC_ID to
 * queues) - action row, observation row and counters of an env then meet in ONE L2, like the env's state rows, and device-scope
 * (sc1) accesses + a wait for the stores are all the ordering needed.  A producer on another XCD should be paired with device-scope
 * fences (a -DGQ_MB_DEBUG build of the library reads them from the environment - GQ_MB_FLAGS: 1 release on the policy side, 2 acquire on
 * the stepping side; measured cost of the latter 14 % - the product build compiles the switches to 0).
 * mode GQ_CLOSED_INLINE: the built-in policy is evaluated by the wavefront that steps the env, right where the observation row is
 * written (the persistent kernel of gq_rollout(shards = 0) with the action derived instead of read): the turn-around of an action
 * is zero instead of two trips through device memory, which matters when there are no more envs than wavefront slots - every env
 * then waits out its own policy latency (measured at 4096 envs: DESIGN.md).  Mailbox mode is the general mechanism: any policy
 * that can run as a resident kernel; its latency hides behind other envs' steps once there are more envs than slots.
 * step_waves: workgroups of the step launch (0 = one per env; more than the device can hold at once, or than there are envs, is harmless:
 * the surplus waits on lap-tagged queue slots or finds its queue drained).
 * obs_seq / act_seq: device [K][N][obs_dim] / [K][N][12] f32 records of every observation row / action, or NULL.
 * Needs the Newton solver, next-step auto-reset (or none) and the production kernel (no inspection record / stage cut).
 * Asynchronous on hip_stream except for one synchronisation at the start (mailbox reset, policy residency). */
typedef struct GqPolicyPd {
  float kp[12], kd[12], q_des[12];   /* hinge order of qpos[7:] */
  float noise_sigma;                 /* Gaussian exploration noise added to every torque: sigma * N(0, 1); 0 = none */
  uint64_t noise_seed;               /* Philox4x32-10 key; counter = (joint, noise_step0 + k, global env id, 0x9011), Box-Muller on words 0, 1 */
  int32_t noise_step0;
} GqPolicyPd;
typedef struct GqMailboxView {
  float* action;            /* device [N][12] */
  int32_t* steps_done;      /* device [N] */
  int32_t* queue_items;     /* device [n_queues][queue_capacity] */
  int32_t* queue_counters;  /* device [n_queues][3][counter_stride]: pop tickets, push tickets, (built-in policy's registration count) */
  int32_t* status;          /* device [8]: word 0 abort code */
  int32_t n_queues, queue_capacity, counter_stride;
  int32_t xcc_queue[16];    /* HW_REG_XCC_ID -> queue: queue q's envs are stepped by wavefronts of that XCD only */
} GqMailboxView;
#define GQ_CLOSED_MAILBOX 0 /* policy = a kernel of its own (built-in PD on a second stream, or the caller's), ready queues */
#define GQ_CLOSED_INLINE 1  /* the stepping wavefront evaluates the built-in policy itself: no mailbox traffic, no policy kernel */
int gq_rollout_closed(GqBatch* b, int n_steps, int mode, const GqPolicyPd* pd, int policy_waves, int step_waves, double timeout_s, GqState st, GqObsOut out,
                      const GqResetCfg* auto_reset, int32_t* episode, uint8_t* lift_failed, float* obs_seq, float* act_seq, void* hip_stream);
/* waits for hip_stream, then out[0] = abort code (0 = the rollout ran to its end), out[1] = who gave up, out[2] = env-steps played;
 * returns GQ_EDEVICE with a message when the rollout was aborted */
int gq_rollout_closed_status(GqBatch* b, int32_t out[4], void* hip_stream);
int gq_mailbox_get(GqBatch* b, GqMailboxView* out);
/* upload the device-resident argument block for (st, out, auto_reset, episode, lift_failed) if it changed; no launch */
int gq_batch_bind(GqBatch* b, GqState st, GqObsOut out, const GqResetCfg* auto_reset, int32_t* episode, uint8_t* lift_failed,
                  void* hip_stream);

/* QuadrupedEnv.reset (quadruped_env.py:309-406) for the envs with mask != 0 (mask NULL = all), two launches:
 *  1. state write: explicit qpos_new/qvel_new when given (:389-391), otherwise keyframe 0 (+ joint noise, random
 *     xy inside terrain_limits, z = hip_height, random roll/pitch, yaw facing the origin when cfg->random, :343-373)
 *     followed by the mj_step1 + "lift until no foot-body contact" loop (:376-388; <= 100 iterations, z += 1.1 *
 *     max|dist|); zero time / qacc / qacc_warmstart / qfrc_applied / step_num (:332-335, :394-395); draw the velocity
 *     command (:400) and the next friction coefficient.
 *  2. one masked mj_step with zero control (:397) that also produces the observation rows (:406); the new
 *     friction coefficient is committed after it (:403-404).
 * episode: device [N] int32 in/out, incremented per reset env (RNG counter).  lift_failed: device [N] u8 out
 * (the reference's RuntimeError condition, :387-388), may be NULL.  qpos_new/qvel_new: device [N][19] f64 /
 * [N][18] f32 or both NULL. */
int gq_reset(GqBatch* b, const uint8_t* mask, const double* qpos_new, const float* qvel_new, const GqResetCfg* cfg,
             GqState st, GqObsOut out, int32_t* episode, uint8_t* lift_failed, void* hip_stream);

/* Checkpoint resume of a next-step auto-reset rollout: overwrite the batch's pending-reset flags with `flags` (device
 * [N] u8; they always equal the `terminated` row of the last gq_step) or clear them (flags NULL).  No reference
 * counterpart - the reference has no auto-reset. */
int gq_batch_set_pending(GqBatch* b, const uint8_t* flags, void* hip_stream);

/* HeightMap.create_sensor_matrix (sensors/heightmap.py:106-169) for every env: rows x cols downward rays
 * (mujoco.mj_ray, heightmap.py:90-99, static geoms only) from the grid centred above `center` and rotated by `yaw`
 * into the heading frame; ray origin z = center.z + 0.6 - 0.07.  center: device [N][3] f64, yaw: device [N] f32,
 * out: device [N][rows][cols][3] f32 hit points ((-1,-1,-1)-style misses cannot occur over the infinite floor). */
int gq_heightmap(GqBatch* b, const double* center, const float* yaw, int rows, int cols, float dist_x, float dist_y,
                 float* out, void* hip_stream);
/* gq_heightmap with row strides (in elements) for `center` and `yaw`: views such as qpos[:, 0:3] (stride 19) or a column of the
 * observation row (stride obs_dim) are read in place - no staging copy on the caller's stream.  yaw_stride 0 = one yaw for all. */
int gq_heightmap_strided(GqBatch* b, const double* center, int center_stride, const float* yaw, int yaw_stride, int rows, int cols,
                         float dist_x, float dist_y, float* out, void* hip_stream);

/* mujoco.mj_jac(m, d, jacp, jacr, point, body) (quadruped_env.py:728-735) for every env: translational and rotational
 * Jacobian of the world point `point` moving with body `body` (MuJoCo body id: 1 = base, 2 + 3 leg + link), at the pose
 * `qpos`.  qpos: device [N][19] f64; point: device [N][3] f64 (world); jacp / jacr: device [N][3][18] f32, either may be
 * NULL (like mj_jac's).  One wavefront per env: kinematics, then lane = dof.  Production kernel, no inspection record. */
int gq_jac(GqBatch* b, const double* qpos, int body, const double* point, float* jacp, float* jacr, void* hip_stream);

/* mujoco.mj_ray(m, d, pnt, vec, geomgroup, flg_static = 1, bodyexclude, geomid) (sensors/heightmap.py:90-99) against the
 * STATIC geoms of the scene - floor plane, world boxes, height field - for n_rays rays per env: origin: device
 * [N][n_rays][3] f64 (world), dir: device [N][n_rays][3] f32 (any length; the result is in units of |dir|, as mj_ray's),
 * dist: device [N][n_rays] f32 out, -1 where nothing is hit; geom: device [N][n_rays] i32 out or NULL (0 floor,
 * 1 + box index, 1 + nbox for the height field, -1 none).  One thread per ray. */
int gq_ray(GqBatch* b, const double* origin, const float* dir, int n_rays, float* dist, int32_t* geom, void* hip_stream);

/* mujoco.mj_step1 (quadruped_env.py:376, :384: position + velocity stages) and mujoco.mj_forward (:1321: through the
 * accelerations) for every env WITHOUT advancing the state: stage 1 = mj_step1 (kinematics, inertias, collision,
 * constraint rows incl. aref; bias forces), stage 0 = mj_forward (+ actuation, solver: st.qacc is written).  qpos / qvel /
 * time / step_num / warm start / observations are left untouched.  The results - what the reference reads from mjData after
 * these calls (xpos, xmat, M, qfrc_bias, contact list, efc_J / aref / R / force, qacc) - land in the inspection record
 * (gq_debug_enable must have been called for the envs of interest; fields: gq_debug_field / gq_debug_device_buffer).  Runs
 * the instrumented kernel variant: ~20 % slower than gq_step's production kernel and one 8.4 KB record per env. */
int gq_forward(GqBatch* b, int stage, const float* ctrl, GqState st, GqObsOut out, void* hip_stream);

/* What the reference reads from mjData AFTER a step for model-based control - mj_fullM (legs_mass_matrix :880-893,
 * get_base_inertia :543-562), qfrc_bias (:895-905), body(i).xpos / xmat (hip_positions :564-595, com :918-929), the foot
 * points mj_jac is evaluated at (:681-740) and the contact list with mj_contactForce (:836-870, :852) - written by the
 * PRODUCTION step kernel as two optional extra rows per env (no instrumented variant, no inspection record):
 *   dyn      device [N][GQ_DYN_STRIDE] f32 or NULL: the tree-sparse joint-space inertia, qfrc_bias, body poses and foot
 *            points of the forward pass (positions with x / y relative to the base x / y of that pass, i.e. qpos before the
 *            step: fp32 never carries the 10 km spawn offsets).  M in mj_fullM's dense form: M[6+j][k<6] = MC[j][k],
 *            M[6+j][6+3*leg(j)+c] = MC[j][6+c] for c <= depth(j) (symmetric; zero between different legs), M[a][b] = MB[a][b].
 *   contacts device [N][GQ_CON_STRIDE] f32 or NULL: word 0 = number of contacts kept (<= GQ_CON_MAX), then one
 *            GQ_CON_REC-float record per contact in MuJoCo's order: geom1 (-1: a world geom - floor, box, height field -
 *            else the robot geom id), geom2 (robot geom id), dist, pos[3] (x / y relative like dyn), frame[9] (rows: normal,
 *            tangent 1, tangent 2 = mjContact.frame), dim, force[6] (mj_contactForce: normal, two tangential, torsional, two
 *            rolling components in the contact frame; zeros beyond dim), friction[0].
 * Both stay registered until changed (NULL, NULL switches them off); cost when on: ~1.3 KB + ~1.2 KB of stores per env-step. */
#define GQ_DYN_MC 0
#define GQ_DYN_MB 108
#define GQ_DYN_BIAS 144
#define GQ_DYN_XPOS 162
#define GQ_DYN_XMAT 201
#define GQ_DYN_FOOT 318
#define GQ_DYN_STRIDE 336
#define GQ_CON_MAX 12
#define GQ_CON_REC 24
#define GQ_CON_STRIDE (8 + GQ_CON_MAX * GQ_CON_REC)
int gq_batch_set_outputs(GqBatch* b, float* dyn, float* contacts);
/* mujoco.mj_contactForce(m, d, id, result) (quadruped_env.py:852) for contact `id` of every env: result device [N][6] f32,
 * zeros for the envs with fewer contacts.  Reads the contact rows registered with gq_batch_set_outputs (a strided device
 * copy, asynchronous on hip_stream). */
int gq_contact_force(GqBatch* b, int id, float* result, void* hip_stream);

/* mujoco.mj_fullM(m, M, d.qM) (quadruped_env.py:557, :884: legs_mass_matrix, get_base_inertia): the dense joint-space
 * inertia of the LAST forward pass (gq_step or gq_forward) of the first n_envs envs, M: device [n_envs][18][18] f32.
 * Like qM in mjData it is a by-product of that pass: it comes out of the inspection record, so gq_debug_enable(b, >= n_envs)
 * must have been called before the pass.  Asynchronous on hip_stream (a strided device-to-device copy, no kernel). */
int gq_full_mass(GqBatch* b, int n_envs, float* M, void* hip_stream);

/* debug / inspection: last forward pass internals of env `env` copied to host (doubles).
 * name in {"M","qfrc_bias","qfrc_smooth","qacc_smooth","qfrc_constraint","efc_J","efc_aref","efc_R","efc_b",
 * "efc_force","efc_type","contact_dist","contact_geom","xpos","xmat","geom_xpos"}; returns count written. */
int gq_debug_enable(GqBatch* b, int enable);
/* The same records where they live: device pointer to the [n_envs][stride] float block the instrumented kernel fills
 * (NULL while disabled), and offset / length of a named field inside one record.  This is what backs the reference's
 * on-demand getters that read mjData after a step - mj_fullM (legs_mass_matrix :881, get_base_inertia :543),
 * qfrc_bias (:895), body(i).xpos (hip_positions :564), mj_jac (feet_jacobians :681) - without a host round trip. */
int gq_debug_device_buffer(GqBatch* b, float** dev, int32_t* n_envs, int32_t* stride);
/* profiling aid (tools/stage_cuts.py): the following gq_step launches return after stage marker `stage` (0 = run
 * everything; the environment variable GQ_STOP_STAGE sets the initial value).  Markers <= 10 have no side effects. */
int gq_debug_stop_stage(GqBatch* b, int stage);
int gq_debug_field(const char* name, int32_t* offset, int32_t* count);
int gq_debug_get(GqBatch* b, int env, const char* name, double* out, int max_n);

#ifdef __cplusplus
}
#endif
#endif /* GQ_H_ */
