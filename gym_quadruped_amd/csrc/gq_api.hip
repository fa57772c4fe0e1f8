/*
 * gq_api.hip - the C-ABI of libgq (include/gq.h): handle management, model upload, launches.
 * No torch types, no hidden synchronisation; every tensor is a caller-owned device pointer.
 */
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "gq.h"
#include "gq_host_model.h"
#include <gq_device.h>
#include "gq_step_kernel.h"
#include "gq_step_body.h"

extern "C" void gq_launch_step(const gq::FusedArgs* dev_args, const gq::StepCall* c, int n_envs, int solver, int cone, int boxes, int self, hipStream_t stream);
extern "C" void gq_launch_reset(const gq::ResetArgs* a, int n_envs, int boxes, hipStream_t stream);
extern "C" void gq_launch_jac(const GqDevModel* model, const double* qpos, int body, const double* point, float* jacp, float* jacr, int n_envs, hipStream_t stream);
extern "C" void gq_launch_ray(const GqDevModel* model, const double* origin, const float* dir, int total, float* dist, int32_t* geom, hipStream_t stream);
extern "C" void gq_launch_heightmap(const GqDevModel* model, const double* center, int center_stride, const float* yaw, int yaw_stride, int n_envs, int rows, int cols,
                                    float dist_x, float dist_y, float* out, hipStream_t stream);

extern "C" void gq_launch_xcc_probe(int32_t* mask, hipStream_t stream);
extern "C" void gq_launch_policy_pd(const gq::MailboxDev* mb, const gq::PolicyPdDev* pd, const float* obs, int od, int waves, hipStream_t stream);
extern "C" int gq_launch_mailbox_step(const gq::FusedArgs* dev_args, const gq::StepCall* c, const gq::MailboxDev* mb, int waves, int solver, int cone, int boxes, int self, hipStream_t stream);

#define GQ_ARG_SLOTS 8
static thread_local char g_err[512] = "";
#define SET_ERR(...) std::snprintf(g_err, sizeof g_err, __VA_ARGS__)
#define HIP_TRY(expr)                                                                   \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess) { SET_ERR("%s: %s", #expr, hipGetErrorString(e_)); return GQ_EDEVICE; } \
  } while (0)

struct GqModel {
  int device;
  GqDevModel host;
  GqDevModel* dev;
  float *vx, *vy, *vz;
  float* hf;            /* device elevations of the scene's height field (NULL: none) */
  int nvert;
};
/* kernel variant by scene: 0 flat, 1 world boxes / height field, 2 the same for a robot with sphere / capsule / box link
 * geoms (exact pair routines compiled in; gq_step_body.h PRIM) */
static int scene_variant(const GqModel* m) {
  if (!(m->host.nbox > 0 || m->host.hf_nrow > 0)) return 0;
  /* lg[] is indexed by link geom (item[] is in contact order: feet and link geoms interleaved by geom id) */
  for (int g = 0; g < m->host.nlg; g++) { const int t = m->host.lg[g].ptype; if (t == 2 || t == 3 || t == 6) return 2; }
  return 1;
}
struct GqBatch {
  GqModel* model;
  GqDevBatch host;
  GqDevBatch* dev;
  float* debug;       /* device, debug_envs * GQ_DBG_SIZE floats (lazily allocated) */
  float* friction_next; /* device [N]: friction drawn by reset, committed after the reset step */
  uint8_t* pending;     /* device [N]: next-step auto-reset flags */
  uint8_t* lift_pending;/* device [N]: reset kernel -> the reset's own step: lift loop still due */
  uint8_t* load_hint;   /* device [N]: per-env solver load of the previous step (scheduling hint of the step kernel) */
  int32_t* xq;          /* device: convex pair exchange (gq_exchange.h) - models with convex self pairs only, else NULL */
  int xq_slots; bool xq_on;
  float* sepc;          /* device: separating-axis cache of the convex self pairs (GqDevBatch::sepc) */
  int stop_stage;       /* profiling aid: GQ_STOP_STAGE at batch creation */
  int force_self;       /* profiling aid: GQ_FORCE_SELF=1 runs the self-collision kernel variant even for a model without pairs */
  /* argument block of step_kernel: device copy, host shadow of what the device holds, pinned staging ring for the
   * (rare) stream-ordered re-upload */
  gq::FusedArgs* dev_args;
  gq::FusedArgs shadow;
  gq::FusedArgs* staging;   /* pinned host, GQ_ARG_SLOTS entries */
  int staging_next;
  /* the batch constants (b->dev): a change made by gq_batch_set_resampling travels with the NEXT launch, on that launch's
   * stream, through its own pinned ring - ordered against everything the caller has queued there */
  GqDevBatch* batch_staging; /* pinned host, GQ_ARG_SLOTS entries */
  int batch_staging_next;
  bool batch_dirty;
  bool shadow_valid;
  float* imu_bias;      /* caller-owned device [N][6], set by gq_batch_set_imu */
  float* heightmap;     /* caller-owned device [N][rows * cols][3], set by gq_batch_set_heightmap (NULL: off) */
  hipStream_t shard_stream[8]; hipEvent_t shard_event[8]; hipEvent_t fork_event; int n_shard_streams; /* gq_rollout */
  int32_t* h9;          /* caller-owned device [N][6] resampling counters, set by gq_batch_set_resampling */
  float* ext_dist;      /* caller-owned device [N][6] */
  int debug_cap;
  float* dyn_out; float* con_out;  /* caller-owned device rows registered with gq_batch_set_outputs */
  /* closed-loop persistent rollout (gq_rollout_closed): mailboxes, ready queues, the policy's stream; allocated on first use */
  struct {
    gq::MailboxDev host;      /* what the device block holds */
    gq::MailboxDev* dev;
    gq::MailboxDev* staging;  /* pinned */
    int32_t* alive;           /* pinned host word the policy workgroups count themselves into */
    int32_t* status_host;     /* pinned copy of the status words (gq_rollout_closed_status) */
    gq::PolicyPdDev* policy_dev; /* device copy of the built-in policy's parameters (inline mode) */
    hipStream_t stream; hipEvent_t fork, join;
    bool ready;
  } mb;
};

/* the launches must be issued with the batch's device current (the caller's stream belongs to it); restore the caller's
 * device afterwards so that a framework sharing the thread is not surprised */
struct DeviceGuard {
  int prev = -1, dev;
  explicit DeviceGuard(int d) : dev(d) { if (hipGetDevice(&prev) == hipSuccess && prev != dev) hipSetDevice(dev); else prev = -1; }
  ~DeviceGuard() { if (prev >= 0) hipSetDevice(prev); }
};

extern "C" {

const char* gq_last_error(void) { return g_err; }
int gq_version(void) { return GQ_ABI_VERSION; }
int gq_struct_sizes(int32_t out[8]) {
  if (!out) { SET_ERR("gq_struct_sizes: null argument"); return GQ_EINVAL; }
  out[0] = (int32_t)sizeof(GqModelDesc); out[1] = (int32_t)sizeof(GqState); out[2] = (int32_t)sizeof(GqObsOut);
  out[3] = (int32_t)sizeof(GqResetCfg); out[4] = (int32_t)sizeof(GqResampleCfg); out[5] = (int32_t)sizeof(GqImuCfg);
  out[6] = (int32_t)sizeof(GqPolicyPd); out[7] = (int32_t)sizeof(GqMailboxView);
  return GQ_OK;
}
int gq_obs_dim(int obs_id) { return gq_obs_dim_host(obs_id); }
int gq_model_destroy(GqModel* m);
int gq_batch_destroy(GqBatch* b);

/* HIP call inside a constructor: on failure the partially built handle is destroyed (frees whatever was allocated) */
#define HIP_TRY_OR_DESTROY(expr, destroy_call)                                          \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess) { SET_ERR("%s: %s", #expr, hipGetErrorString(e_)); destroy_call; return GQ_EDEVICE; } \
  } while (0)

int gq_model_create(const GqModelDesc* desc, int device, GqModel** out) {
  if (!desc || !out) { SET_ERR("gq_model_create: null argument"); return GQ_EINVAL; }
  if (desc->struct_size != (int32_t)sizeof(GqModelDesc)) {
    SET_ERR("gq_model_create: GqModelDesc.struct_size is %d, this library (ABI %d) expects %d - header and library do not match", desc->struct_size, GQ_ABI_VERSION, (int)sizeof(GqModelDesc));
    return GQ_EINVAL;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { SET_ERR("no HIP device visible"); return GQ_ENODEVICE; }
  if (device < 0 || device >= ndev) { SET_ERR("device %d out of range (have %d)", device, ndev); return GQ_EINVAL; }
  GqModel* m = new (std::nothrow) GqModel();   /* value-initialised: every pointer starts NULL */
  if (!m) return GQ_ENOMEM;
  std::vector<float> vx, vy, vz;
  if (gq_build_dev_model(desc, &m->host, &vx, &vy, &vz, g_err, sizeof g_err)) { delete m; return GQ_EINVAL; }
  m->device = device; m->nvert = (int)vx.size();
  DeviceGuard guard(device);
  if (m->host.hf_nrow > 0) {
    std::vector<float> hf;
    gq_hfield_heights(desc, &hf);
    HIP_TRY_OR_DESTROY(hipMalloc(&m->hf, hf.size() * sizeof(float)), gq_model_destroy(m));
    HIP_TRY_OR_DESTROY(hipMemcpy(m->hf, hf.data(), hf.size() * sizeof(float), hipMemcpyHostToDevice), gq_model_destroy(m));
    m->host.hf_data = m->hf;
  }
  HIP_TRY_OR_DESTROY(hipMalloc(&m->dev, sizeof(GqDevModel)), gq_model_destroy(m));
  HIP_TRY_OR_DESTROY(hipMemcpy(m->dev, &m->host, sizeof(GqDevModel), hipMemcpyHostToDevice), gq_model_destroy(m));
  size_t vb = vx.size() * sizeof(float);
  HIP_TRY_OR_DESTROY(hipMalloc(&m->vx, vb), gq_model_destroy(m));
  HIP_TRY_OR_DESTROY(hipMalloc(&m->vy, vb), gq_model_destroy(m));
  HIP_TRY_OR_DESTROY(hipMalloc(&m->vz, vb), gq_model_destroy(m));
  HIP_TRY_OR_DESTROY(hipMemcpy(m->vx, vx.data(), vb, hipMemcpyHostToDevice), gq_model_destroy(m));
  HIP_TRY_OR_DESTROY(hipMemcpy(m->vy, vy.data(), vb, hipMemcpyHostToDevice), gq_model_destroy(m));
  HIP_TRY_OR_DESTROY(hipMemcpy(m->vz, vz.data(), vb, hipMemcpyHostToDevice), gq_model_destroy(m));
  *out = m;
  return GQ_OK;
}

int gq_model_destroy(GqModel* m) {
  if (!m) return GQ_OK;
  DeviceGuard guard(m->device);
  hipFree(m->dev); hipFree(m->vx); hipFree(m->vy); hipFree(m->vz); hipFree(m->hf);
  delete m;
  return GQ_OK;
}

int gq_batch_create(GqModel* m, int n_envs, const int32_t* obs_ids, int n_obs, const int32_t* legs_order, GqBatch** out) {
  if (!m || !out || n_envs <= 0) { SET_ERR("gq_batch_create: bad argument"); return GQ_EINVAL; }
  GqBatch* b = new (std::nothrow) GqBatch();   /* value-initialised: every pointer starts NULL */
  if (!b) return GQ_ENOMEM;
  b->model = m;
  if (gq_build_dev_batch(n_envs, obs_ids, n_obs, legs_order, &b->host, g_err, sizeof g_err)) { delete b; return GQ_EINVAL; }
  DeviceGuard guard(m->device);
  HIP_TRY_OR_DESTROY(hipMalloc(&b->dev, sizeof(GqDevBatch)), gq_batch_destroy(b));
  HIP_TRY_OR_DESTROY(hipMemcpy(b->dev, &b->host, sizeof(GqDevBatch), hipMemcpyHostToDevice), gq_batch_destroy(b));
  HIP_TRY_OR_DESTROY(hipMalloc(&b->friction_next, sizeof(float) * (size_t)n_envs), gq_batch_destroy(b));
  HIP_TRY_OR_DESTROY(hipMemset(b->friction_next, 0, sizeof(float) * (size_t)n_envs), gq_batch_destroy(b));
  HIP_TRY_OR_DESTROY(hipMalloc(&b->pending, (size_t)n_envs), gq_batch_destroy(b));
  HIP_TRY_OR_DESTROY(hipMemset(b->pending, 0, (size_t)n_envs), gq_batch_destroy(b));
  HIP_TRY_OR_DESTROY(hipMalloc(&b->lift_pending, (size_t)n_envs), gq_batch_destroy(b));
  HIP_TRY_OR_DESTROY(hipMemset(b->lift_pending, 0, (size_t)n_envs), gq_batch_destroy(b));
  HIP_TRY_OR_DESTROY(hipMalloc(&b->load_hint, (size_t)n_envs), gq_batch_destroy(b));
  HIP_TRY_OR_DESTROY(hipMemset(b->load_hint, 0, (size_t)n_envs), gq_batch_destroy(b));
  if (m->host.ncvx_self > 0) { /* the pair exchange: two slots per env, rounded up to a power of two (an env publishes what it has beyond its first pair - 0.45 pairs
                                * per env-step on the benchmark's states: the table stays sparse, which is what its hashing wants) */
    int slots = 256;
    while (slots < 2 * n_envs && slots < (1 << 21)) slots <<= 1;
    const size_t words = (size_t)slots * (1 + GQ_XQ_ITEM);
    HIP_TRY_OR_DESTROY(hipMalloc(&b->xq, words * sizeof(int32_t)), gq_batch_destroy(b));
    HIP_TRY_OR_DESTROY(hipMemset(b->xq, 0, words * sizeof(int32_t)), gq_batch_destroy(b));
    b->xq_slots = slots;
    b->xq_on = true;
    b->host.xq = b->xq; b->host.xq_slots = slots;
    HIP_TRY_OR_DESTROY(hipMalloc(&b->sepc, (size_t)n_envs * m->host.ncvx_self * 3 * sizeof(float)), gq_batch_destroy(b));
    HIP_TRY_OR_DESTROY(hipMemset(b->sepc, 0, (size_t)n_envs * m->host.ncvx_self * 3 * sizeof(float)), gq_batch_destroy(b));
    b->host.sepc = b->sepc; b->host.sepc_stride = m->host.ncvx_self * 3;
    HIP_TRY_OR_DESTROY(hipMemcpy(b->dev, &b->host, sizeof(GqDevBatch), hipMemcpyHostToDevice), gq_batch_destroy(b));
  }
  /* profiling knobs of development builds (tools/dev_build.sh defines GQ_DEV_KNOBS; tools/stage_insts.sh, stage_cuts.py): the product library
   * reads no environment variable (tests/test_host_and_abi.py checks its objects for getenv) */
#ifdef GQ_DEV_KNOBS
  { const char* s = getenv("GQ_STOP_STAGE"); b->stop_stage = s ? atoi(s) : 0; }
  { const char* s = getenv("GQ_FORCE_SELF"); b->force_self = (s && atoi(s)) ? 1 : 0; }
#else
  b->stop_stage = 0; b->force_self = 0;
#endif
  HIP_TRY_OR_DESTROY(hipMalloc(&b->dev_args, sizeof(gq::FusedArgs)), gq_batch_destroy(b));
  HIP_TRY_OR_DESTROY(hipHostMalloc(&b->staging, sizeof(gq::FusedArgs) * GQ_ARG_SLOTS, hipHostMallocDefault), gq_batch_destroy(b));
  b->staging_next = 0; b->shadow_valid = false;
  HIP_TRY_OR_DESTROY(hipHostMalloc(&b->batch_staging, sizeof(GqDevBatch) * GQ_ARG_SLOTS, hipHostMallocDefault), gq_batch_destroy(b));
  b->batch_staging_next = 0; b->batch_dirty = false;
  std::memset(&b->shadow, 0, sizeof b->shadow);
  *out = b;
  return GQ_OK;
}

/* releases whatever mailbox_setup has allocated so far (every pointer of the block starts out null: GqBatch is value-initialised) */
static void mailbox_free(GqBatch* b) {
  auto& m = b->mb;
  hipFree(m.host.act); hipFree(m.host.steps_done); hipFree(m.host.issued); hipFree(m.host.q_items); hipFree(m.host.q_ctr); hipFree(m.host.status);
  hipFree(m.dev); hipFree(m.policy_dev);
  if (m.staging) hipHostFree(m.staging);
  if (m.alive) hipHostFree(m.alive);
  if (m.status_host) hipHostFree(m.status_host);
  if (m.stream) hipStreamDestroy(m.stream);
  if (m.fork) hipEventDestroy(m.fork);
  if (m.join) hipEventDestroy(m.join);
  std::memset(&m, 0, sizeof m);
}

int gq_batch_destroy(GqBatch* b) {
  if (!b) return GQ_OK;
  DeviceGuard guard(b->model->device);
  hipFree(b->dev); hipFree(b->friction_next); hipFree(b->pending); hipFree(b->lift_pending); hipFree(b->load_hint); hipFree(b->xq); hipFree(b->sepc); hipFree(b->dev_args);
  if (b->staging) hipHostFree(b->staging);
  if (b->batch_staging) hipHostFree(b->batch_staging);
  mailbox_free(b);
  for (int i = 0; i < b->n_shard_streams; i++) { hipStreamDestroy(b->shard_stream[i]); hipEventDestroy(b->shard_event[i]); }
  if (b->n_shard_streams) hipEventDestroy(b->fork_event);
  if (b->debug) hipFree(b->debug);
  delete b;
  return GQ_OK;
}

int gq_batch_obs_dim(const GqBatch* b) { return b ? b->host.obs_dim : GQ_EINVAL; }

int gq_batch_set_imu(GqBatch* b, const GqImuCfg* cfg, float* bias_state) {
  if (!b || !cfg || !bias_state) { SET_ERR("gq_batch_set_imu: null argument"); return GQ_EINVAL; }
  gq_fill_imu(&b->host, cfg);
  b->imu_bias = bias_state;
  b->batch_dirty = true; /* uploaded by the next launch, stream-ordered (ensure_args) */
  return GQ_OK;
}

int gq_batch_set_pair_exchange(GqBatch* b, int on) {
  if (!b) { SET_ERR("gq_batch_set_pair_exchange: null batch"); return GQ_EINVAL; }
  if (on && !b->xq) { SET_ERR("gq_batch_set_pair_exchange: the model has no convex self pairs - nothing to exchange"); return GQ_EINVAL; }
  b->xq_on = on != 0;
  b->host.xq = b->xq_on ? b->xq : nullptr; b->host.xq_slots = b->xq_on ? b->xq_slots : 0;
  b->batch_dirty = true; /* uploaded by the next launch, stream-ordered (ensure_args) */
  return GQ_OK;
}

int gq_batch_set_heightmap(GqBatch* b, int rows, int cols, float dist_x, float dist_y, float* out) {
  if (!b) { SET_ERR("gq_batch_set_heightmap: null batch"); return GQ_EINVAL; }
  if (!out) { b->heightmap = nullptr; b->host.hm_rows = b->host.hm_cols = 0; b->batch_dirty = true; return GQ_OK; }
  if (rows <= 0 || cols <= 0 || rows > 4096 || cols > 4096 || rows * cols > 4096 || !(dist_x > 0.0f) || !(dist_y > 0.0f)) { SET_ERR("gq_batch_set_heightmap: bad grid (%d x %d cells of %g x %g m)", rows, cols, (double)dist_x, (double)dist_y); return GQ_EINVAL; }
  if (scene_variant(b->model) == 0) { SET_ERR("gq_batch_set_heightmap: the scene has no world boxes / height field - every ray ends on the floor plane; use gq_heightmap"); return GQ_EINVAL; }
  b->heightmap = out;
  b->host.hm_rows = rows; b->host.hm_cols = cols; b->host.hm_dx = dist_x; b->host.hm_dy = dist_y;
  b->batch_dirty = true; /* uploaded by the next launch, stream-ordered (ensure_args) */
  return GQ_OK;
}

int gq_batch_set_resampling(GqBatch* b, const GqResampleCfg* cfg, const GqResetCfg* cmd_cfg, int32_t* counters, float* ext_dist) {
  if (!b) { SET_ERR("gq_batch_set_resampling: null batch"); return GQ_EINVAL; }
  GqDevBatch& h = b->host;
  if (!cfg) { h.rs_cmd_reset = 0; h.rs_dist_reset = 0; b->h9 = nullptr; b->ext_dist = nullptr; }
  else {
    if (!counters || ((cfg->cmd_reset != 0) && !cmd_cfg) || ((cfg->dist_reset != 0) && !ext_dist)) {
      SET_ERR("gq_batch_set_resampling: counters (and the command knobs / wrench tensor of the enabled parts) are required"); return GQ_EINVAL;
    }
    h.rs_cmd_reset = cfg->cmd_reset != 0; h.rs_dist_reset = cfg->dist_reset != 0; h.rs_env_id_offset = cfg->env_id_offset;
    for (int k = 0; k < 6; k++) { h.rs_dist_kind[k] = cfg->dist_kind[k]; h.rs_dist_range[k][0] = cfg->dist_range[k][0]; h.rs_dist_range[k][1] = cfg->dist_range[k][1]; }
    if (cmd_cfg) {
      for (int k = 0; k < 2; k++) { h.rs_lin_vel_range[k] = cmd_cfg->lin_vel_range[k]; h.rs_ang_vel_range[k] = cmd_cfg->ang_vel_range[k]; }
      h.rs_cmd_forward = cmd_cfg->cmd_forward; h.rs_cmd_random = cmd_cfg->cmd_random; h.rs_cmd_rotate = cmd_cfg->cmd_rotate;
    }
    h.rs_seed_lo = (uint32_t)(cfg->seed & 0xffffffffu); h.rs_seed_hi = (uint32_t)(cfg->seed >> 32);
    b->h9 = counters; b->ext_dist = ext_dist;
  }
  b->batch_dirty = true; /* uploaded by the next launch, stream-ordered (ensure_args) */
  return GQ_OK;
}

int gq_batch_set_outputs(GqBatch* b, float* dyn, float* contacts) {
  if (!b) { SET_ERR("gq_batch_set_outputs: null batch"); return GQ_EINVAL; }
  b->dyn_out = dyn; b->con_out = contacts;   /* picked up by the next launch's argument block (ensure_args) */
  return GQ_OK;
}

int gq_contact_force(GqBatch* b, int id, float* result, void* hip_stream) {
  if (!b || !result) { SET_ERR("gq_contact_force: null argument"); return GQ_EINVAL; }
  if (!b->con_out) { SET_ERR("gq_contact_force: no contact rows registered (gq_batch_set_outputs)"); return GQ_EINVAL; }
  if (id < 0 || id >= GQ_CON_MAX) { SET_ERR("gq_contact_force: contact id %d out of range (0..%d)", id, GQ_CON_MAX - 1); return GQ_EINVAL; }
  DeviceGuard guard(b->model->device);
  /* records past an env's contact count are written as zeros by the kernel */
  HIP_TRY(hipMemcpy2DAsync(result, 6 * sizeof(float), b->con_out + 8 + id * GQ_CON_REC + 16, GQ_CON_STRIDE * sizeof(float), 6 * sizeof(float),
                           (size_t)b->host.n_envs, hipMemcpyDeviceToDevice, (hipStream_t)hip_stream));
  return GQ_OK;
}

int gq_debug_enable(GqBatch* b, int n_debug_envs) {
  if (!b) return GQ_EINVAL;
  if (n_debug_envs > b->host.n_envs) n_debug_envs = b->host.n_envs;
  DeviceGuard guard(b->model->device);
  if (n_debug_envs > b->debug_cap) {
    if (b->debug) hipFree(b->debug);
    HIP_TRY(hipMalloc(&b->debug, (size_t)n_debug_envs * GQ_DBG_SIZE * sizeof(float)));
    HIP_TRY(hipMemset(b->debug, 0, (size_t)n_debug_envs * GQ_DBG_SIZE * sizeof(float)));
    b->debug_cap = n_debug_envs;
  }
  b->host.debug_envs = n_debug_envs;
  b->batch_dirty = true; /* uploaded by the next launch, stream-ordered (ensure_args) */
  return GQ_OK;
}

static void fill_reset_cfg(gq::ResetCfgDev* d, const GqResetCfg* cfg) {
  d->seed_lo = (uint32_t)(cfg->seed & 0xffffffffu); d->seed_hi = (uint32_t)(cfg->seed >> 32);
  d->random = cfg->random; d->q_pos_amp = cfg->q_pos_amp; d->q_vel_amp = cfg->q_vel_amp;
  d->roll_sweep = cfg->roll_sweep; d->pitch_sweep = cfg->pitch_sweep; d->hip_height = cfg->hip_height;
  for (int k = 0; k < 2; k++) { d->lin_vel_range[k] = cfg->lin_vel_range[k]; d->ang_vel_range[k] = cfg->ang_vel_range[k]; d->friction_range[k] = cfg->friction_range[k]; }
  d->cmd_forward = cfg->cmd_forward; d->cmd_random = cfg->cmd_random; d->cmd_rotate = cfg->cmd_rotate; d->cmd_human = cfg->cmd_human;
  d->env_id_offset = cfg->env_id_offset;
}
static void fill_reset_args(gq::ResetArgs* a, GqBatch* b, const uint8_t* mask, const double* qpos_new, const float* qvel_new,
                            const GqResetCfg* cfg, const GqState& st, const GqObsOut& out, int32_t* episode, uint8_t* lift_failed);
static void fill_step_args(gq::StepArgs* a, GqBatch* b, const GqState& st, const GqObsOut& out, const int32_t* episode, uint8_t* lift_failed) {
  GqModel* m = b->model;
  a->model = m->dev; a->batch = b->dev; a->vx = m->vx; a->vy = m->vy; a->vz = m->vz;
  a->qpos = st.qpos; a->qvel = st.qvel; a->qacc = st.qacc; a->warm = st.qacc_warmstart;
  a->applied = st.qfrc_applied; a->time = st.time; a->friction = st.friction; a->cmd = st.cmd;
  a->friction_next = b->friction_next; a->pending = b->pending; a->load_hint = b->load_hint;
  a->imu_bias = b->imu_bias; a->heightmap = b->heightmap;
  a->h9 = b->h9; a->ext_dist = b->ext_dist;
  a->dyn = b->dyn_out; a->contacts = b->con_out;
  a->lift_failed = lift_failed; a->lift_pending = b->lift_pending;
  a->episode_ro = episode;
  a->obs = out.obs; a->reward = out.reward; a->terminated = out.terminated; a->truncated = out.truncated;
  a->invalid_contact = out.invalid_contact; a->step_num = out.step_num; a->step_prev = out.step_num_prev;
  a->contacts_dropped = out.contacts_dropped;
  a->n_envs = b->host.n_envs;
  a->timestep = m->host.timestep; a->nlg = m->host.nlg; a->nfl = m->host.nfl; a->pad_ = 0;
}
/* Make the device argument block describe (st, out, episode, lift_failed[, auto-reset cfg]).  Steady state: a memcmp.
 * On a change the new block goes through a pinned staging slot with a stream-ordered copy, so launches already queued
 * on `stream` still see the old block.  reset_cfg NULL keeps whatever auto-reset block the device holds. */
/* batch constants changed since the last launch (gq_batch_set_resampling / _set_imu / gq_debug_enable): stream-ordered upload */
static int flush_batch(GqBatch* b, hipStream_t stream) {
  if (!b->batch_dirty) return GQ_OK;
  if (b->batch_staging_next == GQ_ARG_SLOTS) { HIP_TRY(hipStreamSynchronize(stream)); b->batch_staging_next = 0; }
  GqDevBatch* slot = b->batch_staging + b->batch_staging_next++;
  std::memcpy(slot, &b->host, sizeof(GqDevBatch));
  HIP_TRY(hipMemcpyAsync(b->dev, slot, sizeof(GqDevBatch), hipMemcpyHostToDevice, stream));
  b->batch_dirty = false;
  return GQ_OK;
}
static int ensure_args(GqBatch* b, const GqState& st, const GqObsOut& out, int32_t* episode, uint8_t* lift_failed,
                       const GqResetCfg* reset_cfg, hipStream_t stream) {
  { const int rcb = flush_batch(b, stream); if (rcb != GQ_OK) return rcb; }
  gq::FusedArgs want = b->shadow; /* struct copy keeps padding bytes identical for the memcmp */
  fill_step_args(&want.s, b, st, out, episode, lift_failed);
  if (reset_cfg) fill_reset_args(&want.r, b, nullptr, nullptr, nullptr, reset_cfg, st, out, episode, lift_failed);
  if (b->shadow_valid && std::memcmp(&want, &b->shadow, sizeof want) == 0) return GQ_OK;
  if (b->staging_next == GQ_ARG_SLOTS) { /* every slot may still be in flight: drain before reusing the ring */
    HIP_TRY(hipStreamSynchronize(stream));
    b->staging_next = 0;
  }
  gq::FusedArgs* slot = b->staging + b->staging_next++;
  std::memcpy(slot, &want, sizeof want);
  HIP_TRY(hipMemcpyAsync(b->dev_args, slot, sizeof want, hipMemcpyHostToDevice, stream));
  b->shadow = want; b->shadow_valid = true;
  return GQ_OK;
}
static void fill_reset_args(gq::ResetArgs* a, GqBatch* b, const uint8_t* mask, const double* qpos_new, const float* qvel_new,
                            const GqResetCfg* cfg, const GqState& st, const GqObsOut& out, int32_t* episode, uint8_t* lift_failed) {
  GqModel* m = b->model;
  a->model = m->dev; a->vx = m->vx; a->vy = m->vy; a->vz = m->vz; a->mask = mask; a->qpos_new = qpos_new; a->qvel_new = qvel_new;
  a->qpos = st.qpos; a->qvel = st.qvel; a->qacc = st.qacc; a->warm = st.qacc_warmstart; a->applied = st.qfrc_applied;
  a->time = st.time; a->cmd = st.cmd; a->friction_next = st.friction ? b->friction_next : nullptr;
  a->step_num = out.step_num; a->episode = episode; a->lift_failed = lift_failed;
  a->h9 = b->h9;
  a->lift_pending = nullptr;   /* fused auto-reset: the wave hands the flag to its own step; gq_reset sets the scratch pointer */
  fill_reset_cfg(&a->cfg, cfg);
  a->cfg.cmd_reset = b->host.rs_cmd_reset;
}

static int step_launch(GqBatch* b, int env0, int count, const float* ctrl, const uint8_t* mask, GqState st, GqObsOut out, const GqResetCfg* auto_reset,
                       int32_t* episode, uint8_t* lift_failed, void* hip_stream, const char* who) {
  if (!b || !st.qpos || !st.qvel || !st.qacc || !st.qacc_warmstart || !st.time || !out.obs || !out.reward ||
      !out.terminated || !out.truncated || !out.invalid_contact || !out.step_num) {
    SET_ERR("%s: null tensor", who); return GQ_EINVAL;
  }
  if (env0 < 0 || count < 0 || env0 + count > b->host.n_envs) { SET_ERR("%s: env range [%d, %d) outside the batch of %d", who, env0, env0 + count, b->host.n_envs); return GQ_EINVAL; }
  DeviceGuard guard(b->model->device);
  if (auto_reset && (!episode || !st.cmd)) { SET_ERR("%s: auto-reset needs the episode counters and the command tensor", who); return GQ_EINVAL; }
  if (b->host.rs_cmd_reset && !st.cmd) { SET_ERR("%s: command resampling is on (gq_batch_set_resampling) but the state has no command tensor", who); return GQ_EINVAL; }
  const int rc = ensure_args(b, st, out, episode, lift_failed, auto_reset, (hipStream_t)hip_stream);
  if (rc != GQ_OK) return rc;
  if (count == 0) return GQ_OK;
  gq::StepCall c{};
  c.ctrl = ctrl; c.mask = mask; c.debug = b->host.debug_envs > 0 ? b->debug : nullptr; c.env0 = env0;
  c.auto_reset = auto_reset ? (auto_reset->autoreset_next_step ? 2 : 1) : 0; c.first_pass = 0; c.stop_stage = b->stop_stage;
  gq_launch_step(b->dev_args, &c, count, b->model->host.solver, b->model->host.cone, scene_variant(b->model), (b->model->host.nsp > 0 || b->force_self), (hipStream_t)hip_stream);
  HIP_TRY(hipGetLastError());
  return GQ_OK;
}

int gq_step(GqBatch* b, const float* ctrl, const uint8_t* mask, GqState st, GqObsOut out, const GqResetCfg* auto_reset,
            int32_t* episode, uint8_t* lift_failed, void* hip_stream) {
  return step_launch(b, 0, b ? b->host.n_envs : 0, ctrl, mask, st, out, auto_reset, episode, lift_failed, hip_stream, "gq_step");
}

int gq_step_range(GqBatch* b, int env0, int count, const float* ctrl, GqState st, GqObsOut out, const GqResetCfg* auto_reset,
                  int32_t* episode, uint8_t* lift_failed, void* hip_stream) {
  return step_launch(b, env0, count, ctrl, nullptr, st, out, auto_reset, episode, lift_failed, hip_stream, "gq_step_range");
}

int gq_rollout(GqBatch* b, const float* ctrl_seq, int n_steps, int shards, GqState st, GqObsOut out, const GqResetCfg* auto_reset,
               int32_t* episode, uint8_t* lift_failed, float* obs_seq, void* hip_stream) {
  if (!b || !ctrl_seq || n_steps < 0) { SET_ERR("gq_rollout: bad argument"); return GQ_EINVAL; }
  /* the persistent kernel exists for the production variant only: with the inspection record or a stage cut active the rollout is
   * played as the step loop (one shard), so that the record describes the last step and the cut applies to every step */
  if (shards == 0 && b && (b->host.debug_envs > 0 || b->stop_stage != 0)) shards = 1;
  if (shards == 0) { /* persistent: ONE launch, every wavefront plays the whole sequence of its env (StepCall::n_steps) */
    if (auto_reset && !auto_reset->autoreset_next_step) { SET_ERR("gq_rollout: the persistent rollout (shards = 0) needs next-step auto-reset or none"); return GQ_EINVAL; }
    int rc0 = step_launch(b, 0, 0, nullptr, nullptr, st, out, auto_reset, episode, lift_failed, hip_stream, "gq_rollout"); /* validate + bind */
    if (rc0 != GQ_OK) return rc0;
    if (n_steps == 0) return GQ_OK;
    DeviceGuard guard0(b->model->device);
    gq::StepCall c{};
    c.ctrl = ctrl_seq; c.n_steps = n_steps; c.ctrl_stride = b->host.n_envs * 12; c.obs_seq = obs_seq;
    c.auto_reset = auto_reset ? 2 : 0; c.stop_stage = b->stop_stage;
    gq_launch_step(b->dev_args, &c, b->host.n_envs, b->model->host.solver, b->model->host.cone, scene_variant(b->model),
                   (b->model->host.nsp > 0 || b->force_self), (hipStream_t)hip_stream);
    HIP_TRY(hipGetLastError());
    return GQ_OK;
  }
  if (shards < 1) shards = 1;
  if (shards > 8) shards = 8;
  if (shards > b->host.n_envs) shards = b->host.n_envs;
  int rc = step_launch(b, 0, 0, nullptr, nullptr, st, out, auto_reset, episode, lift_failed, hip_stream, "gq_rollout"); /* validate + bind */
  if (rc != GQ_OK) return rc;
  DeviceGuard guard(b->model->device);
  while (b->n_shard_streams < shards) {
    const int i = b->n_shard_streams;
    if (i == 0) HIP_TRY(hipEventCreateWithFlags(&b->fork_event, hipEventDisableTiming));
    HIP_TRY(hipStreamCreateWithFlags(&b->shard_stream[i], hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&b->shard_event[i], hipEventDisableTiming));
    b->n_shard_streams = i + 1;
  }
  const int N = b->host.n_envs, od = b->host.obs_dim;
  HIP_TRY(hipEventRecord(b->fork_event, (hipStream_t)hip_stream));
  /* from here on work may sit on the library's shard streams: whatever fails, the caller's stream is made to wait for them
   * before the error is returned - the caller may free ctrl_seq / obs_seq as soon as ITS stream is done with them */
  hipError_t herr = hipSuccess;
  const char* what = "";
#define RO_TRY(x) do { if (herr == hipSuccess) { herr = (x); if (herr != hipSuccess) what = #x; } } while (0)
  for (int s = 0; s < shards; s++) RO_TRY(hipStreamWaitEvent(b->shard_stream[s], b->fork_event, 0));
  gq::StepCall c{};
  c.debug = b->host.debug_envs > 0 ? b->debug : nullptr;
  c.auto_reset = auto_reset ? (auto_reset->autoreset_next_step ? 2 : 1) : 0; c.stop_stage = b->stop_stage;
  for (int k = 0; k < n_steps && herr == hipSuccess; k++) {
    c.ctrl = ctrl_seq + (size_t)k * N * 12;
    for (int s = 0; s < shards && herr == hipSuccess; s++) {
      const int e0 = (int)((long long)s * N / shards), e1 = (int)((long long)(s + 1) * N / shards);
      c.env0 = e0;
      gq_launch_step(b->dev_args, &c, e1 - e0, b->model->host.solver, b->model->host.cone, scene_variant(b->model),
                     (b->model->host.nsp > 0 || b->force_self), b->shard_stream[s]);
      RO_TRY(hipGetLastError());
      if (obs_seq) RO_TRY(hipMemcpyAsync(obs_seq + ((size_t)k * N + e0) * od, out.obs + (size_t)e0 * od, (size_t)(e1 - e0) * od * sizeof(float), hipMemcpyDeviceToDevice, b->shard_stream[s]));
    }
  }
  for (int s = 0; s < shards; s++) { /* join - also on the error path (a stream that cannot even record its event is drained on the host) */
    hipError_t j = hipEventRecord(b->shard_event[s], b->shard_stream[s]);
    if (j == hipSuccess) j = hipStreamWaitEvent((hipStream_t)hip_stream, b->shard_event[s], 0);
    if (j != hipSuccess) { (void)hipStreamSynchronize(b->shard_stream[s]); if (herr == hipSuccess) { herr = j; what = "joining the shard streams"; } }
  }
#undef RO_TRY
  if (herr != hipSuccess) { SET_ERR("gq_rollout: HIP error '%s' at %s", hipGetErrorString(herr), what); return GQ_EDEVICE; }
  return GQ_OK;
}

/* mailboxes, queues and the policy stream of a batch; the XCD census of the device (one probe launch) */
static int mailbox_setup_impl(GqBatch* b);
static int mailbox_setup(GqBatch* b) {
  if (b->mb.ready) return GQ_OK;
  const int rc = mailbox_setup_impl(b);
  if (rc != GQ_OK) mailbox_free(b); /* a failed setup keeps nothing: the next call starts from scratch */
  return rc;
}
static int mailbox_setup_impl(GqBatch* b) {
  const int N = b->host.n_envs;
  gq::MailboxDev& h = b->mb.host;
  std::memset(&h, 0, sizeof h);
  int qcap = 64;
  while (qcap < N) qcap <<= 1;
  /* which XCC ids do the workgroups of this device report?  (8 on an MI355X in SPX mode; a partitioned device shows fewer) */
  int32_t* mask_dev = nullptr;
  HIP_TRY(hipMalloc(&mask_dev, sizeof(int32_t)));
  HIP_TRY(hipMemset(mask_dev, 0, sizeof(int32_t)));
  gq_launch_xcc_probe(mask_dev, 0);
  int32_t mask = 0;
  HIP_TRY(hipMemcpy(&mask, mask_dev, sizeof mask, hipMemcpyDeviceToHost));
  hipFree(mask_dev);
  if (mask == 0) { SET_ERR("gq_rollout_closed: the XCD probe saw no workgroup"); return GQ_EDEVICE; }
  int nq = 0;
  for (int x = 0; x < 16; x++) h.xcc_queue[x] = ((mask >> x) & 1) ? nq++ : 0;
  h.nq = nq; h.qcap = qcap; h.n_envs = N;
  HIP_TRY(hipMalloc(&h.act, sizeof(float) * 12 * (size_t)N));
  HIP_TRY(hipMalloc(&h.steps_done, sizeof(int32_t) * (size_t)N));
  HIP_TRY(hipMalloc(&h.issued, sizeof(int32_t) * 2 * (size_t)N)); /* + N words of the XCD census experiment */
  HIP_TRY(hipMalloc(&h.q_items, sizeof(int32_t) * (size_t)nq * qcap));
  HIP_TRY(hipMalloc(&h.q_ctr, sizeof(int32_t) * (size_t)nq * 3 * GQ_MB_QSTRIDE));
  HIP_TRY(hipMalloc(&h.status, sizeof(int32_t) * 8));
  HIP_TRY(hipMemset(h.status, 0, sizeof(int32_t) * 8));
  HIP_TRY(hipMalloc(&b->mb.dev, sizeof(gq::MailboxDev)));
  HIP_TRY(hipMalloc(&b->mb.policy_dev, sizeof(gq::PolicyPdDev)));
  HIP_TRY(hipHostMalloc(&b->mb.staging, sizeof(gq::MailboxDev), hipHostMallocDefault));
  HIP_TRY(hipHostMalloc(&b->mb.alive, sizeof(int32_t), hipHostMallocDefault));
  HIP_TRY(hipHostMalloc(&b->mb.status_host, sizeof(int32_t) * 8, hipHostMallocDefault));
  h.alive = b->mb.alive;
  HIP_TRY(hipStreamCreateWithFlags(&b->mb.stream, hipStreamNonBlocking));
  HIP_TRY(hipEventCreateWithFlags(&b->mb.fork, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&b->mb.join, hipEventDisableTiming));
  b->mb.ready = true;
  return GQ_OK;
}

#ifdef GQ_MB_DEBUG
/* experiment hook of -DGQ_MB_DEBUG builds (tools/closed_loop_debug.py; not part of the ABI): copy the XCD census words of the last closed rollout to the host */
int gq_mailbox_census(GqBatch* b, int32_t* out_host) {
  if (!b || !b->mb.ready) return GQ_EINVAL;
  DeviceGuard guard(b->model->device);
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out_host, b->mb.host.issued + b->host.n_envs, sizeof(int32_t) * (size_t)b->host.n_envs, hipMemcpyDeviceToHost));
  return GQ_OK;
}
#endif

int gq_mailbox_get(GqBatch* b, GqMailboxView* out) {
  if (!b || !out) { SET_ERR("gq_mailbox_get: null argument"); return GQ_EINVAL; }
  DeviceGuard guard(b->model->device);
  const int rc = mailbox_setup(b);
  if (rc != GQ_OK) return rc;
  const gq::MailboxDev& h = b->mb.host;
  out->action = h.act; out->steps_done = h.steps_done; out->queue_items = h.q_items; out->queue_counters = h.q_ctr; out->status = h.status;
  out->n_queues = h.nq; out->queue_capacity = h.qcap; out->counter_stride = GQ_MB_QSTRIDE;
  for (int x = 0; x < 16; x++) out->xcc_queue[x] = h.xcc_queue[x];
  return GQ_OK;
}

int gq_rollout_closed(GqBatch* b, int n_steps, int mode, const GqPolicyPd* pd, int policy_waves, int step_waves, double timeout_s, GqState st, GqObsOut out,
                      const GqResetCfg* auto_reset, int32_t* episode, uint8_t* lift_failed, float* obs_seq, float* act_seq, void* hip_stream) {
  if (!b || n_steps < 0 || (mode != GQ_CLOSED_MAILBOX && mode != GQ_CLOSED_INLINE)) { SET_ERR("gq_rollout_closed: bad argument"); return GQ_EINVAL; }
  if (mode == GQ_CLOSED_INLINE && !pd) { SET_ERR("gq_rollout_closed: the inline mode runs the built-in policy: pd must be given"); return GQ_EINVAL; }
  if (auto_reset && !auto_reset->autoreset_next_step) { SET_ERR("gq_rollout_closed needs next-step auto-reset or none"); return GQ_EINVAL; }
  if (b->model->host.solver != 1) { SET_ERR("gq_rollout_closed needs the Newton solver (solver = 1)"); return GQ_EINVAL; }
  if (b->host.debug_envs > 0 || b->stop_stage != 0) { SET_ERR("gq_rollout_closed runs the production kernel: switch the inspection record / stage cut off first"); return GQ_EINVAL; }
  hipStream_t stream = (hipStream_t)hip_stream;
  int rc = step_launch(b, 0, 0, nullptr, nullptr, st, out, auto_reset, episode, lift_failed, hip_stream, "gq_rollout_closed"); /* validate + bind */
  if (rc != GQ_OK) return rc;
  DeviceGuard guard(b->model->device);
  rc = mailbox_setup(b);
  if (rc != GQ_OK) return rc;
  if (n_steps == 0) return GQ_OK;
  const int N = b->host.n_envs, od = b->host.obs_dim;
  gq::MailboxDev& h = b->mb.host;
  /* queue tickets are 32-bit: the env-steps that pass through one queue must stay below 2^31 (items carry env + 1 in 24 bits) */
  if (N >= (1 << 24) || (int64_t)((N + h.nq - 1) / h.nq) * (int64_t)n_steps >= ((int64_t)1 << 31) - 65536) {
    SET_ERR("gq_rollout_closed: %d envs x %d steps over %d queues overflows the 32-bit ticket counters: split the rollout", N, n_steps, h.nq); return GQ_EINVAL;
  }
  gq::PolicyPdDev P{};
  if (pd) { /* the columns of the joint angles / velocities in this batch's observation row */
    for (int j = 0; j < 12; j++) {
      P.kp[j] = pd->kp[j]; P.kd[j] = pd->kd[j]; P.qdes[j] = pd->q_des[j]; P.col_q[j] = -1; P.col_qd[j] = -1;
      for (int c = 0; c < od; c++) {
        const int src = b->host.obs_map[c];
        if (P.col_q[j] < 0 && (src == gq::OB_QPOS_JS + j || src == gq::OB_QPOS + 7 + j)) P.col_q[j] = c;
        if (P.col_qd[j] < 0 && (src == gq::OB_QVEL_JS + j || src == gq::OB_QVEL + 6 + j)) P.col_qd[j] = c;
      }
      if (P.col_q[j] < 0 || P.col_qd[j] < 0) { SET_ERR("gq_rollout_closed: the PD policy reads qpos_js / qvel_js (or qpos / qvel): not in this batch's observation row"); return GQ_EINVAL; }
    }
    if (policy_waves <= 0) policy_waves = 64; /* lane = env: 4096 envs get a lane each */
    if (policy_waves > 256) policy_waves = 256;
    if (policy_waves < 2 * h.nq) policy_waves = 2 * h.nq; /* every XCD needs policy wavefronts of its own (dispatch is round-robin over the XCDs) */
    P.sigma = pd->noise_sigma; P.seed_lo = (uint32_t)(pd->noise_seed & 0xffffffffu); P.seed_hi = (uint32_t)(pd->noise_seed >> 32);
    P.step0 = pd->noise_step0; P.env_id_offset = auto_reset ? auto_reset->env_id_offset : 0;
    /* the parameter block lives in device memory for both modes */
    if (b->staging_next == GQ_ARG_SLOTS) { HIP_TRY(hipStreamSynchronize(stream)); b->staging_next = 0; }
    gq::PolicyPdDev* slot = reinterpret_cast<gq::PolicyPdDev*>(b->staging + b->staging_next++); /* a pinned staging slot of the argument ring */
    std::memcpy(slot, &P, sizeof P);
    HIP_TRY(hipMemcpyAsync(b->mb.policy_dev, slot, sizeof P, hipMemcpyHostToDevice, stream));
  }
  if (mode == GQ_CLOSED_INLINE) {
    /* the persistent rollout kernel with the policy evaluated by the stepping wavefront itself: no mailbox, no second kernel */
    HIP_TRY(hipMemsetAsync(h.status, 0, sizeof(int32_t) * 8, stream));
    gq::StepCall ci{};
    ci.n_steps = n_steps; ci.obs_seq = obs_seq; ci.act_seq = act_seq; ci.policy = b->mb.policy_dev;
    ci.auto_reset = auto_reset ? 2 : 0; ci.stop_stage = 0;
    gq_launch_step(b->dev_args, &ci, N, b->model->host.solver, b->model->host.cone, scene_variant(b->model), (b->model->host.nsp > 0 || b->force_self), stream);
    HIP_TRY(hipGetLastError());
    return GQ_OK;
  }
  if (step_waves <= 0) step_waves = N; /* more workgroups than free slots (or than envs) is harmless: pop tickets that run ahead of the pushes
                                        * wait on lap-tagged slots, the late ones find the queues drained */
  if (step_waves < 4 * h.nq) step_waves = 4 * h.nq; /* a wavefront pops from ITS XCD's queue only: every XCD needs stepping wavefronts, also for a batch of
                                                      * one env (workgroups are dealt round-robin over the XCDs; the surplus finds its queue empty and leaves) */
  h.n_steps = n_steps; h.obs_seq = obs_seq; h.act_seq = act_seq;
  h.timeout_ticks = (int64_t)((timeout_s > 0.0 ? timeout_s : 5.0) * 1e8);
  h.flags = 0;
#ifdef GQ_MB_DEBUG
#ifdef GQ_DEV_KNOBS
  { const char* fl = getenv("GQ_MB_FLAGS"); h.flags = fl ? atoi(fl) : 0; } /* fence / census experiments (tools/closed_loop_debug.py), development builds only */
#else
  h.flags = 0;
#endif
#endif
  /* fresh rollout state, ordered on the caller's stream */
  HIP_TRY(hipMemsetAsync(h.steps_done, 0, sizeof(int32_t) * (size_t)N, stream));
  HIP_TRY(hipMemsetAsync(h.issued, 0, sizeof(int32_t) * 2 * (size_t)N, stream));
  HIP_TRY(hipMemsetAsync(h.q_items, 0, sizeof(int32_t) * (size_t)h.nq * h.qcap, stream));
  HIP_TRY(hipMemsetAsync(h.q_ctr, 0, sizeof(int32_t) * (size_t)h.nq * 3 * GQ_MB_QSTRIDE, stream));
  HIP_TRY(hipMemsetAsync(h.status, 0, sizeof(int32_t) * 8, stream));
  HIP_TRY(hipStreamSynchronize(stream)); /* the pinned staging block below is reused per call; a rollout is thousands of launches' worth of work */
  std::memcpy(b->mb.staging, &h, sizeof h);
  HIP_TRY(hipMemcpyAsync(b->mb.dev, b->mb.staging, sizeof h, hipMemcpyHostToDevice, stream));
  gq::StepCall c{};
  c.auto_reset = auto_reset ? 2 : 0;
  const int boxes = scene_variant(b->model), self = (b->model->host.nsp > 0 || b->force_self);
  if (pd) {
    /* the policy must be RESIDENT before the step wavefronts take every slot of the device: launch it first, on its own stream,
     * and wait until each of its workgroups has reported in */
    *b->mb.alive = 0;
    HIP_TRY(hipEventRecord(b->mb.fork, stream));
    HIP_TRY(hipStreamWaitEvent(b->mb.stream, b->mb.fork, 0));
    gq_launch_policy_pd(b->mb.dev, b->mb.policy_dev, out.obs, od, policy_waves, b->mb.stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(b->mb.join, b->mb.stream));
    const auto t0 = std::chrono::steady_clock::now();
    while (*(volatile int32_t*)b->mb.alive < policy_waves) {
      std::this_thread::yield();
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2.0) {
        /* it cannot start: tell it to leave as soon as it does, and report */
        b->mb.status_host[0] = 3; /* pinned: the source of an asynchronous copy must outlive this frame */
        (void)hipMemcpyAsync(h.status, b->mb.status_host, sizeof(int32_t), hipMemcpyHostToDevice, stream);
        (void)hipStreamSynchronize(stream);
        (void)hipStreamSynchronize(b->mb.stream); /* the policy kernel has left (or never ran): nothing of this call touches `alive` later */
        SET_ERR("gq_rollout_closed: the policy kernel did not become resident within 2 s (%d of %d workgroups)", (int)*(volatile int32_t*)b->mb.alive, policy_waves);
        return GQ_EDEVICE;
      }
    }
  }
  if (!gq_launch_mailbox_step(b->dev_args, &c, b->mb.dev, step_waves, b->model->host.solver, b->model->host.cone, boxes, self, stream)) {
    SET_ERR("gq_rollout_closed: no mailbox variant of the step kernel for this model in this build"); return GQ_EINVAL;
  }
  HIP_TRY(hipGetLastError());
  if (pd) HIP_TRY(hipStreamWaitEvent(stream, b->mb.join, 0)); /* the caller's stream resumes when both kernels are done */
  return GQ_OK;
}

int gq_rollout_closed_status(GqBatch* b, int32_t out[4], void* hip_stream) {
  if (!b || !out) { SET_ERR("gq_rollout_closed_status: null argument"); return GQ_EINVAL; }
  if (!b->mb.ready) { out[0] = out[1] = out[2] = out[3] = 0; return GQ_OK; }
  DeviceGuard guard(b->model->device);
  HIP_TRY(hipMemcpyAsync(b->mb.status_host, b->mb.host.status, sizeof(int32_t) * 8, hipMemcpyDeviceToHost, (hipStream_t)hip_stream));
  HIP_TRY(hipStreamSynchronize((hipStream_t)hip_stream));
  for (int i = 0; i < 4; i++) out[i] = b->mb.status_host[i];
  if (out[0] != 0) { SET_ERR("closed-loop rollout aborted: code %d (1: a step wavefront waited past the deadline for ticket %d; 2: policy lane %d waited past the deadline; 3: policy not resident; 4: no policy wavefront on XCD queue %d), %d env-steps were played", out[0], out[1], out[1], out[1], out[2]); return GQ_EDEVICE; }
  return GQ_OK;
}

int gq_batch_bind(GqBatch* b, GqState st, GqObsOut out, const GqResetCfg* auto_reset, int32_t* episode, uint8_t* lift_failed, void* hip_stream) {
  return step_launch(b, 0, 0, nullptr, nullptr, st, out, auto_reset, episode, lift_failed, hip_stream, "gq_batch_bind");
}

int gq_reset(GqBatch* b, const uint8_t* mask, const double* qpos_new, const float* qvel_new, const GqResetCfg* cfg,
             GqState st, GqObsOut out, int32_t* episode, uint8_t* lift_failed, void* hip_stream) {
  if (!b || !cfg || !st.qpos || !st.qvel || !st.qacc || !st.qacc_warmstart || !st.time || !out.step_num || !out.obs ||
      !out.reward || !out.terminated || !out.truncated || !out.invalid_contact) {
    SET_ERR("gq_reset: null tensor"); return GQ_EINVAL;
  }
  if ((qpos_new == nullptr) != (qvel_new == nullptr)) { SET_ERR("gq_reset: qpos_new and qvel_new must be given together"); return GQ_EINVAL; }
  DeviceGuard guard(b->model->device);
  gq::ResetArgs r{};
  fill_reset_args(&r, b, mask, qpos_new, qvel_new, cfg, st, out, episode, lift_failed);
  r.clear_terminated = out.terminated; r.clear_truncated = out.truncated; r.clear_invalid = out.invalid_contact;
  r.lift_pending = b->lift_pending;
  { const int rcb = flush_batch(b, (hipStream_t)hip_stream); if (rcb != GQ_OK) return rcb; }
  gq_launch_reset(&r, b->host.n_envs, scene_variant(b->model), (hipStream_t)hip_stream);
  HIP_TRY(hipGetLastError());
  /* the reset's own mj_step with zero control (quadruped_env.py:334, :397); friction committed after it (:403-404) */
  const int rc = ensure_args(b, st, out, episode, lift_failed, nullptr, (hipStream_t)hip_stream);
  if (rc != GQ_OK) return rc;
  gq::StepCall c{};
  c.mask = mask; c.first_pass = 1; c.debug = b->host.debug_envs > 0 ? b->debug : nullptr;
  gq_launch_step(b->dev_args, &c, b->host.n_envs, b->model->host.solver, b->model->host.cone, scene_variant(b->model), (b->model->host.nsp > 0 || b->force_self), (hipStream_t)hip_stream);
  HIP_TRY(hipGetLastError());
  return GQ_OK;
}

int gq_batch_set_pending(GqBatch* b, const uint8_t* flags, void* hip_stream) {
  if (!b) { SET_ERR("gq_batch_set_pending: null batch"); return GQ_EINVAL; }
  DeviceGuard guard(b->model->device);
  if (flags) HIP_TRY(hipMemcpyAsync(b->pending, flags, (size_t)b->host.n_envs, hipMemcpyDeviceToDevice, (hipStream_t)hip_stream));
  else HIP_TRY(hipMemsetAsync(b->pending, 0, (size_t)b->host.n_envs, (hipStream_t)hip_stream));
  return GQ_OK;
}

int gq_heightmap(GqBatch* b, const double* center, const float* yaw, int rows, int cols, float dist_x, float dist_y,
                 float* out, void* hip_stream) {
  if (!b || !center || !yaw || !out || rows <= 0 || cols <= 0) { SET_ERR("gq_heightmap: bad argument"); return GQ_EINVAL; }
  DeviceGuard guard(b->model->device);
  gq_launch_heightmap(b->model->dev, center, 3, yaw, 1, b->host.n_envs, rows, cols, dist_x, dist_y, out, (hipStream_t)hip_stream);
  HIP_TRY(hipGetLastError());
  return GQ_OK;
}

int gq_heightmap_strided(GqBatch* b, const double* center, int center_stride, const float* yaw, int yaw_stride, int rows, int cols, float dist_x,
                         float dist_y, float* out, void* hip_stream) {
  if (!b || !center || !yaw || !out || rows <= 0 || cols <= 0 || center_stride < 0 || yaw_stride < 0) { SET_ERR("gq_heightmap_strided: bad argument"); return GQ_EINVAL; }
  DeviceGuard guard(b->model->device);
  gq_launch_heightmap(b->model->dev, center, center_stride, yaw, yaw_stride, b->host.n_envs, rows, cols, dist_x, dist_y, out, (hipStream_t)hip_stream);
  HIP_TRY(hipGetLastError());
  return GQ_OK;
}

int gq_jac(GqBatch* b, const double* qpos, int body, const double* point, float* jacp, float* jacr, void* hip_stream) {
  if (!b || !qpos || !point || (!jacp && !jacr)) { SET_ERR("gq_jac: null argument"); return GQ_EINVAL; }
  if (body < 1 || body > GQ_NB) { SET_ERR("gq_jac: body id %d out of range (1 = base .. %d)", body, GQ_NB); return GQ_EINVAL; }
  DeviceGuard guard(b->model->device);
  gq_launch_jac(b->model->dev, qpos, body, point, jacp, jacr, b->host.n_envs, (hipStream_t)hip_stream);
  HIP_TRY(hipGetLastError());
  return GQ_OK;
}

int gq_ray(GqBatch* b, const double* origin, const float* dir, int n_rays, float* dist, int32_t* geom, void* hip_stream) {
  if (!b || !origin || !dir || !dist || n_rays <= 0) { SET_ERR("gq_ray: bad argument"); return GQ_EINVAL; }
  DeviceGuard guard(b->model->device);
  gq_launch_ray(b->model->dev, origin, dir, b->host.n_envs * n_rays, dist, geom, (hipStream_t)hip_stream);
  HIP_TRY(hipGetLastError());
  return GQ_OK;
}

int gq_forward(GqBatch* b, int stage, const float* ctrl, GqState st, GqObsOut out, void* hip_stream) {
  if (!b || !st.qpos || !st.qvel || !st.qacc || !st.qacc_warmstart || !st.time || !out.obs || !out.reward || !out.terminated ||
      !out.truncated || !out.invalid_contact || !out.step_num) { SET_ERR("gq_forward: null tensor"); return GQ_EINVAL; }
  if (stage != 0 && stage != 1) { SET_ERR("gq_forward: stage must be 0 (mj_forward) or 1 (mj_step1)"); return GQ_EINVAL; }
  if (b->host.debug_envs <= 0 || !b->debug) { SET_ERR("gq_forward: no inspection record to write to (call gq_debug_enable first)"); return GQ_EINVAL; }
  if (b->model->host.solver != 1) { SET_ERR("gq_forward needs the Newton solver (solver = 1)"); return GQ_EINVAL; }
  DeviceGuard guard(b->model->device);
  const int rc = ensure_args(b, st, out, nullptr, nullptr, nullptr, (hipStream_t)hip_stream);
  if (rc != GQ_OK) return rc;
  gq::StepCall c{};
  c.ctrl = ctrl; c.debug = b->debug; c.forward = stage == 1 ? 1 : 2;
  gq_launch_step(b->dev_args, &c, b->host.n_envs, b->model->host.solver, b->model->host.cone, scene_variant(b->model),
                 (b->model->host.nsp > 0 || b->force_self), (hipStream_t)hip_stream);
  HIP_TRY(hipGetLastError());
  return GQ_OK;
}

static const struct { const char* name; int off, n; } kDbg[] = {
    {"M", GQ_DBG_M, 324}, {"qfrc_bias", GQ_DBG_BIAS, 18}, {"qfrc_smooth", GQ_DBG_SMOOTH, 18},
    {"qacc_smooth", GQ_DBG_QACC_SMOOTH, 18}, {"qfrc_constraint", GQ_DBG_QFRC_C, 18}, {"xpos", GQ_DBG_XPOS, 39},
    {"xmat", GQ_DBG_XMAT, 117}, {"nefc", GQ_DBG_NEFC, 1}, {"ncon", GQ_DBG_NCON, 1}, {"niter", GQ_DBG_NITER, 1},
    {"efc_J", GQ_DBG_EFC_J, 64 * 18}, {"efc_aref", GQ_DBG_EFC_AREF, 64}, {"efc_R", GQ_DBG_EFC_R, 64},
    {"efc_b", GQ_DBG_EFC_B, 64}, {"efc_force", GQ_DBG_EFC_FORCE, 64}, {"efc_type", GQ_DBG_EFC_TYPE, 64},
    {"contact_dist", GQ_DBG_CON_DIST, GQ_MAXCON}, {"contact_geom", GQ_DBG_CON_GEOM, GQ_MAXCON},
    {"foot_pos", GQ_DBG_FOOT_POS, 12}, {"qacc", GQ_DBG_QACC, 18}, {"timer", GQ_DBG_TIMER, 32}, {"xq", GQ_DBG_XQ, 16}, {"record", 0, GQ_DBG_SIZE}};

int gq_debug_stop_stage(GqBatch* b, int stage) {
  if (!b) { SET_ERR("gq_debug_stop_stage: null batch"); return GQ_EINVAL; }
  b->stop_stage = stage;
  return GQ_OK;
}

int gq_debug_field(const char* name, int32_t* offset, int32_t* count) {
  if (!name || !offset || !count) { SET_ERR("gq_debug_field: null argument"); return GQ_EINVAL; }
  for (const auto& f : kDbg)
    if (!std::strcmp(f.name, name)) { *offset = f.off; *count = f.n; return GQ_OK; }
  SET_ERR("gq_debug_field: unknown field %s", name);
  return GQ_EINVAL;
}

int gq_debug_device_buffer(GqBatch* b, float** dev, int32_t* n_envs, int32_t* stride) {
  if (!b || !dev || !n_envs || !stride) { SET_ERR("gq_debug_device_buffer: null argument"); return GQ_EINVAL; }
  *dev = b->host.debug_envs > 0 ? b->debug : nullptr; *n_envs = b->host.debug_envs; *stride = GQ_DBG_SIZE;
  return GQ_OK;
}

int gq_full_mass(GqBatch* b, int n_envs, float* M, void* hip_stream) {
  if (!b || !M || n_envs <= 0) { SET_ERR("gq_full_mass: null / empty argument"); return GQ_EINVAL; }
  if (!b->debug || n_envs > b->host.debug_envs) {
    SET_ERR("gq_full_mass: the inspection record covers %d envs, %d requested (gq_debug_enable first, then gq_step / gq_forward)", b->host.debug_envs, n_envs);
    return GQ_EINVAL;
  }
  DeviceGuard guard(b->model->device);
  HIP_TRY(hipMemcpy2DAsync(M, 324 * sizeof(float), b->debug + GQ_DBG_M, GQ_DBG_SIZE * sizeof(float), 324 * sizeof(float), (size_t)n_envs,
                           hipMemcpyDeviceToDevice, (hipStream_t)hip_stream));
  return GQ_OK;
}

int gq_debug_get(GqBatch* b, int env, const char* name, double* out, int max_n) {
  if (!b || !name || !out || env < 0 || env >= b->host.debug_envs || !b->debug) { SET_ERR("gq_debug_get: bad argument / debug not enabled"); return GQ_EINVAL; }
  for (const auto& f : kDbg)
    if (!std::strcmp(f.name, name)) {
      int n = f.n < max_n ? f.n : max_n;
      std::vector<float> tmp((size_t)n);
      DeviceGuard guard(b->model->device);
      HIP_TRY(hipDeviceSynchronize());
      HIP_TRY(hipMemcpy(tmp.data(), b->debug + (size_t)env * GQ_DBG_SIZE + f.off, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
      for (int i = 0; i < n; i++) out[i] = tmp[(size_t)i];
      return n;
    }
  SET_ERR("gq_debug_get: unknown field %s", name);
  return GQ_EINVAL;
}

}  // extern "C"
