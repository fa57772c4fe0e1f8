/*
 * gq_kernels.hip - gfx950 kernels of libgq: one environment per 64-lane wavefront, one wavefront per workgroup
 * (grid = n_envs).  With 4096 envs and <= 128 VGPRs / <= 10 KB LDS per wave the whole batch is resident at once
 * (4 waves per SIMD, 16 per CU); workgroup b lands on XCD b % 8, so consecutive envs spread over all eight L2s and
 * every XCD keeps its own copy of the read-only model block.
 */
#include <gq_device.h>
#include "gq_step_body.h"

namespace gq {

/* step (+ in-kernel auto-reset).  Same-step mode: a terminated env is re-spawned by the same wavefront - reset_wave,
 * then the reset's own mj_step as a second pass through step_wave; no extra launches, but the launch lasts as long as
 * its two-pass waves.  Next-step mode: the env waits (pending flag) and spends its next launch on reset_wave + the
 * reset's mj_step instead of a user step - every wave runs exactly one mj_step per launch. */
template <int SOLVER, int MODE, bool CONE, bool BOXES, bool SELF>
__global__ void __launch_bounds__(GQ_WAVE, 4) step_kernel(const FusedArgs* __restrict__ A, const StepCall c) {
  if (c.mask && !gptr(c.mask)[blockIdx.x]) return; /* wave-uniform */
  __shared__ WaveMem W;
  int pass = c.first_pass;
  bool respawn = c.auto_reset == 2 && gptr(A->s.pending)[blockIdx.x]; /* wave-uniform */
  /* the reset's own step after an explicit gq_reset: the reset kernel left word whether the lift loop is still due */
  int lift = (c.first_pass && A->s.lift_pending) ? (int)gptr(A->s.lift_pending)[blockIdx.x] : 0;
  for (;;) { /* one call site each for reset_wave / step_wave: both are large and fully inlined */
    if (respawn) {
      wave_priority(3); /* reset + step in one launch: this wave is the longest of its SIMD */
      lift = reset_wave<BOXES>(A->r, W);
      pass = c.auto_reset;
    }
    const int term = step_wave<SOLVER, MODE, CONE, BOXES, SELF>(A->s, c, W, pass, lift);
    if (pass != 0 || c.auto_reset != 1 || !term) break;
    respawn = true;
  }
}

template <bool BOXES>
__global__ void __launch_bounds__(GQ_WAVE) reset_kernel(ResetArgs a) {
  if (a.mask && !gptr(a.mask)[blockIdx.x]) return;
  __shared__ WaveMem W;
  reset_wave<BOXES>(a, W);
}

/* HeightMap rays: one thread per (env, cell).  Scene = the floor plane z = 0 plus the static world boxes (mj_ray against
 * static geoms, heightmap.py:90-99): the nearest hit of the vertical ray with any box (slab test in the box frame). */
__global__ void heightmap_kernel(const GQ_GLOBAL GqDevModel* model, const double* center, const float* yaw, int n_envs, int rows, int cols,
                                 float dist_x, float dist_y, float* out) {
  const int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x), cells = rows * cols;
  if (idx >= n_envs * cells) return;
  const int env = idx / cells, cell = idx % cells, i = cell / cols, j = cell % cols;
  const float c_rows = (rows % 2 == 0) ? 0.5f * rows : 0.5f * (rows - 1), c_cols = (cols % 2 == 0) ? 0.5f * cols : 0.5f * (cols - 1);
  const float off_r = (rows % 2 == 0) ? -0.5f * dist_x : 0.0f, off_c = (cols % 2 == 0) ? -0.5f * dist_y : 0.0f;
  const float ox = dist_x * (c_rows - (float)i) + off_r, oy = dist_y * (c_cols - (float)j) + off_c;
  const float cy = cosf(yaw[env]), sy = sinf(yaw[env]);
  /* offset in the world frame: R_W2H^T [ox, oy], R_W2H = [[c, s], [-s, c]] */
  const double px = center[(size_t)env * 3 + 0] + (double)(cy * ox - sy * oy);
  const double py = center[(size_t)env * 3 + 1] + (double)(sy * ox + cy * oy);
  const double pz = center[(size_t)env * 3 + 2] + 0.6 - 0.07;
  /* mj_ray along -z against the floor plane: distance = pz (ray starts above the floor), hit = origin - z * dist */
  double dist = pz > 0.0 ? pz : -1.0;   /* mj_ray returns -1 when nothing is hit */
  const int nbox = model->nbox;
  for (int b = 0; b < nbox; b++) {
    const GQ_GLOBAL GqDevBox& B = model->box[b];
    const double ox = px - (double)B.pos[0], oy = py - (double)B.pos[1], oz = pz - (double)B.pos[2];
    if (ox * ox + oy * oy > (double)(B.rad * B.rad)) continue; /* the vertical ray misses the bounding sphere */
    /* origin and direction (0, 0, -1) in the box frame */
    double tin = 0.0, tout = 1e30;
    bool hit = true;
    for (int k = 0; k < 3 && hit; k++) {
      const double ol = (double)B.mat[k] * ox + (double)B.mat[3 + k] * oy + (double)B.mat[6 + k] * oz, dl = -(double)B.mat[6 + k];
      const double s = (double)B.size[k];
      if (fabs(dl) < 1e-12) { hit = fabs(ol) <= s; continue; }
      double t0 = (-s - ol) / dl, t1 = (s - ol) / dl;
      if (t0 > t1) { const double tt = t0; t0 = t1; t1 = tt; }
      if (t0 > tin) tin = t0;
      if (t1 < tout) tout = t1;
      hit = tin <= tout;
    }
    if (hit && tout >= 0.0 && (dist < 0.0 || tin < dist)) dist = tin;
  }
  if (model->hf_nrow > 0) { /* height field: the vertical ray meets the triangle under (px, py) */
    const GQ_GLOBAL GqDevModel& M = *model;
    const float x = (float)(px - (double)M.hf_pos[0]), y = (float)(py - (double)M.hf_pos[1]);
    const float fx = (x + M.hf_sx) * M.hf_inv_dx, fy = (y + M.hf_sy) * M.hf_inv_dy;
    if (fx >= 0.0f && fy >= 0.0f && fx <= (float)(M.hf_ncol - 1) && fy <= (float)(M.hf_nrow - 1)) {
      const int nc = M.hf_ncol, c = min((int)fx, nc - 2), r = min((int)fy, M.hf_nrow - 2);
      const float u = fx - (float)c, v = fy - (float)r;
      const float* H = M.hf_data;
      const float h00 = H[r * nc + c], h10 = H[r * nc + c + 1], h01 = H[(r + 1) * nc + c], h11 = H[(r + 1) * nc + c + 1];
      const float h = u + v <= 1.0f ? h00 + u * (h10 - h00) + v * (h01 - h00) : h11 + (1.0f - u) * (h01 - h11) + (1.0f - v) * (h10 - h11);
      const double top = (double)M.hf_pos[2] + (double)h, t = pz - top;
      if (t >= 0.0 && (dist < 0.0 || t < dist)) dist = t;
    }
  }
  float* o = out + (size_t)idx * 3;
  o[0] = (float)px; o[1] = (float)py; o[2] = (float)(pz - dist);
}

}  // namespace gq

extern "C" void gq_launch_heightmap(const GQ_GLOBAL GqDevModel* model, const double* center, const float* yaw, int n_envs, int rows, int cols,
                                    float dist_x, float dist_y, float* out, hipStream_t stream) {
  const int total = n_envs * rows * cols;
  hipLaunchKernelGGL(gq::heightmap_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, model, center, yaw, n_envs, rows, cols,
                     dist_x, dist_y, out);
}

extern "C" void gq_launch_step(const gq::FusedArgs* dev_args, const gq::StepCall* c, int n_envs, int solver, int cone, int boxes, int self, hipStream_t stream) {
  /* 0: production; 1: debug record + stage timers; 2: stage cut (GQ_STOP_STAGE / gq_debug_stop_stage) - the early returns
   * of the cut cost the production kernel ~8 % when merely compiled in, hence a variant of their own.
   * Scene variants: flat (no world geoms beyond the floor), flat + robot self-collision, world boxes / height field (always
   * with the self-collision stage compiled in; a model without pairs skips it at run time). */
  const int mode = c->debug != nullptr ? 1 : (c->stop_stage != 0 ? 2 : 0);
#define GQ_LAUNCH(S, M, C) do { if (boxes) hipLaunchKernelGGL((gq::step_kernel<S, M, C, true, true>), dim3(n_envs), dim3(GQ_WAVE), 0, stream, dev_args, *c); \
                                else if (self) hipLaunchKernelGGL((gq::step_kernel<S, M, C, false, true>), dim3(n_envs), dim3(GQ_WAVE), 0, stream, dev_args, *c); \
                                else hipLaunchKernelGGL((gq::step_kernel<S, M, C, false, false>), dim3(n_envs), dim3(GQ_WAVE), 0, stream, dev_args, *c); } while (0)
#define GQ_LAUNCH_MODE(S, C) do { if (mode == 1) GQ_LAUNCH(S, 1, C); else if (mode == 2) GQ_LAUNCH(S, 2, C); else GQ_LAUNCH(S, 0, C); } while (0)
  if (solver == 1 && cone) GQ_LAUNCH_MODE(1, true);
  else if (solver == 1) GQ_LAUNCH_MODE(1, false);
  else { /* PGS: floor plane only (gq_model_create rejects world boxes / self-collision with solver 0) */
    if (mode == 1) hipLaunchKernelGGL((gq::step_kernel<0, 1, false, false, false>), dim3(n_envs), dim3(GQ_WAVE), 0, stream, dev_args, *c);
    else if (mode == 2) hipLaunchKernelGGL((gq::step_kernel<0, 2, false, false, false>), dim3(n_envs), dim3(GQ_WAVE), 0, stream, dev_args, *c);
    else hipLaunchKernelGGL((gq::step_kernel<0, 0, false, false, false>), dim3(n_envs), dim3(GQ_WAVE), 0, stream, dev_args, *c);
  }
#undef GQ_LAUNCH_MODE
#undef GQ_LAUNCH
}
extern "C" void gq_launch_reset(const gq::ResetArgs* a, int n_envs, int boxes, hipStream_t stream) {
  if (boxes) hipLaunchKernelGGL(gq::reset_kernel<true>, dim3(n_envs), dim3(GQ_WAVE), 0, stream, *a);
  else hipLaunchKernelGGL(gq::reset_kernel<false>, dim3(n_envs), dim3(GQ_WAVE), 0, stream, *a);
}
