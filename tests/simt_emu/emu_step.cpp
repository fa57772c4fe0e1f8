/* TEST INFRASTRUCTURE - runs the unmodified step kernel body (csrc/gq_step_body.h) under the host SIMT emulator.
 * Exposes a C entry point with the same tensors as gq_step, operating on host memory. */
#include <functional>
#include <vector>
#include <cstdio>

#include "gq_device.h"          /* the emulator shim (this directory comes first on the include path) */
#include "gq_step_body.h"
#include "gq_host_model.h"

void emu_run_wave(unsigned block, unsigned nblocks, const std::function<void()>& body);

static void emu_fill_cfg(gq::ResetCfgDev* d, const GqResetCfg* cfg) {
  d->seed_lo = (uint32_t)(cfg->seed & 0xffffffffu); d->seed_hi = (uint32_t)(cfg->seed >> 32);
  d->random = cfg->random; d->q_pos_amp = cfg->q_pos_amp; d->q_vel_amp = cfg->q_vel_amp;
  d->roll_sweep = cfg->roll_sweep; d->pitch_sweep = cfg->pitch_sweep; d->hip_height = cfg->hip_height;
  for (int k = 0; k < 2; k++) { d->lin_vel_range[k] = cfg->lin_vel_range[k]; d->ang_vel_range[k] = cfg->ang_vel_range[k]; d->friction_range[k] = cfg->friction_range[k]; }
  d->cmd_forward = cfg->cmd_forward; d->cmd_random = cfg->cmd_random; d->cmd_rotate = cfg->cmd_rotate; d->cmd_human = cfg->cmd_human;
  d->env_id_offset = cfg->env_id_offset;
}

/* the convex pair exchange (gq_exchange.h) under emulation: wavefronts run one after the other, so an owner ends up claiming its own items -
 * every queue operation is exercised, concurrency is not (tests/test_gpu_parity.py compares exchange on / off on the device) */
static int g_emu_xq_on = 0;
extern "C" void emu_set_exchange(int on) { g_emu_xq_on = on; }
extern "C" void emu_exchange_stats(int* out) { out[0] = gq::g_emu_cas_ok; out[1] = out[2] = out[3] = 0; }

/* same control flow as gq::step_kernel (csrc/gq_kernels.hip) */
extern "C" int emu_step(const GqModelDesc* desc, int n_envs, const int32_t* obs_ids, int n_obs, const int32_t* legs_order,
                        const float* ctrl, const uint8_t* mask, double* qpos, float* qvel, float* qacc, float* warm,
                        float* applied, float* time, float* friction, float* cmd, float* obs,
                        float* reward, uint8_t* terminated, uint8_t* truncated, uint8_t* invalid_contact,
                        int32_t* step_num, float* debug, int debug_envs, const GqResetCfg* auto_reset, int32_t* episode,
                        uint8_t* lift_failed, float* friction_next, int first_pass, const GqImuCfg* imu, float* imu_bias,
                        uint8_t* pending, uint8_t* lift_pending, char* err, int errlen) {
  static GqDevModel M;
  static GqDevBatch B;
  std::vector<float> vx, vy, vz;
  if (gq_build_dev_model(desc, &M, &vx, &vy, &vz, err, (size_t)errlen)) return -1;
  static std::vector<float> hf_heights;
  gq_hfield_heights(desc, &hf_heights);
  M.hf_data = hf_heights.empty() ? nullptr : hf_heights.data();
  if (gq_build_dev_batch(n_envs, obs_ids, n_obs, legs_order, &B, err, (size_t)errlen)) return -1;
  B.debug_envs = debug_envs;
  if (imu) gq_fill_imu(&B, imu);
  gq::FusedArgs f{};
  gq::StepArgs& a = f.s;
  a.model = &M; a.batch = &B; a.vx = vx.data(); a.vy = vy.data(); a.vz = vz.data();
  a.qpos = qpos; a.qvel = qvel; a.qacc = qacc; a.warm = warm; a.applied = applied;
  a.time = time; a.friction = friction; a.cmd = cmd; a.friction_next = friction_next; a.pending = pending; a.obs = obs; a.reward = reward;
  a.terminated = terminated; a.truncated = truncated; a.invalid_contact = invalid_contact; a.step_num = step_num;
  a.n_envs = n_envs; a.imu_bias = imu ? imu_bias : nullptr; a.episode_ro = episode;
  a.lift_failed = lift_failed; a.lift_pending = lift_pending;
  a.timestep = M.timestep; a.nlg = M.nlg; a.nfl = M.nfl;
  gq::StepCall call{};
  call.ctrl = ctrl; call.mask = mask; call.debug = debug;
  call.auto_reset = auto_reset ? (auto_reset->autoreset_next_step ? 2 : 1) : 0; call.first_pass = first_pass;
  static std::vector<int32_t> xq;
  static std::vector<float> sepc; /* the separating-axis cache (GqDevBatch::sepc): kept across calls like the batch's own */
  if (M.ncvx_self > 0) {
    if (sepc.size() != (size_t)n_envs * M.ncvx_self * 3) sepc.assign((size_t)n_envs * M.ncvx_self * 3, 0.0f);
    B.sepc = sepc.data(); B.sepc_stride = M.ncvx_self * 3;
  }
  if (g_emu_xq_on && M.ncvx_self > 0) {
    const int slots = 256;
    if (xq.empty()) xq.assign((size_t)slots * (1 + GQ_XQ_ITEM), 0);
    B.xq = xq.data(); B.xq_slots = slots;
  }
  if (auto_reset) {
    gq::ResetArgs& r = f.r;
    r.model = &M; r.vx = vx.data(); r.vy = vy.data(); r.vz = vz.data();
    r.qpos = qpos; r.qvel = qvel; r.qacc = qacc; r.warm = warm; r.applied = applied; r.time = time; r.cmd = cmd;
    r.friction_next = friction_next; r.step_num = step_num; r.episode = episode; r.lift_failed = lift_failed;
    emu_fill_cfg(&r.cfg, auto_reset);
  }
  for (int e = 0; e < n_envs; e++) {
    if (mask && !mask[e]) continue;
    emu_run_wave((unsigned)e, (unsigned)n_envs, [&]() {
      __shared__ gq::WaveMem W;
      int pass = call.first_pass;
      gq::WaveCtx C;
      const bool boxes = M.nbox > 0 || M.hf_nrow > 0, self = M.nsp > 0;
      bool prims = false; /* as gq_api.hip scene_variant: the PRIM variants serve robots with sphere / capsule / box link geoms */
      for (int g = 0; g < M.nlg; g++) prims = prims || M.lg[g].ptype == 2 || M.lg[g].ptype == 3 || M.lg[g].ptype == 6;
      int hint = M.solver == 1 ? gq::load_rows<1>(f.s, call, W, e, pass == 0, C) : gq::load_rows<0>(f.s, call, W, e, pass == 0, C);
      bool respawn = call.auto_reset == 2 && C.pend;
      for (;;) {
        if (respawn) {
          gq::wave_barrier();
          boxes ? (prims ? gq::reset_wave<true, true>(f.r, W) : gq::reset_wave<true, false>(f.r, W)) : gq::reset_wave<false>(f.r, W);
          pass = call.auto_reset;
          hint = M.solver == 1 ? gq::load_rows<1>(f.s, call, W, e, false, C, true) : gq::load_rows<0>(f.s, call, W, e, false, C, true);
        }
        int term;
        if (M.solver != 1) { /* PGS (pyramidal cones only) */
          if (boxes && prims) term = gq::step_wave<0, 1, false, true, true, true>(f.s, call, W, pass, hint, C);
          else if (boxes) term = gq::step_wave<0, 1, false, true, true, false>(f.s, call, W, pass, hint, C);
          else if (self) term = gq::step_wave<0, 1, false, false, true, true>(f.s, call, W, pass, hint, C);
          else term = gq::step_wave<0, 1, false, false, false, true>(f.s, call, W, pass, hint, C);
        }
        else if (boxes && prims) term = M.cone ? gq::step_wave<1, 1, true, true, true, true>(f.s, call, W, pass, hint, C) : gq::step_wave<1, 1, false, true, true, true>(f.s, call, W, pass, hint, C);
        else if (boxes) term = M.cone ? gq::step_wave<1, 1, true, true, true, false>(f.s, call, W, pass, hint, C) : gq::step_wave<1, 1, false, true, true, false>(f.s, call, W, pass, hint, C);
        else if (self) term = M.cone ? gq::step_wave<1, 1, true, false, true, true>(f.s, call, W, pass, hint, C) : gq::step_wave<1, 1, false, false, true, true>(f.s, call, W, pass, hint, C);
        else term = M.cone ? gq::step_wave<1, 1, true, false, false, true>(f.s, call, W, pass, hint, C) : gq::step_wave<1, 1, false, false, false, true>(f.s, call, W, pass, hint, C);
        if (pass != 0 || call.auto_reset != 1 || !term) break;
        respawn = true;
      }
    });
  }
  return B.obs_dim;
}

extern "C" int emu_reset(const GqModelDesc* desc, int n_envs, const uint8_t* mask, const double* qpos_new, const float* qvel_new,
                         const GqResetCfg* cfg, double* qpos, float* qvel, float* qacc, float* warm, float* applied,
                         float* time, float* cmd, float* friction_next, int32_t* step_num, int32_t* episode,
                         uint8_t* lift_failed, uint8_t* lift_pending, char* err, int errlen) {
  static GqDevModel M;
  std::vector<float> vx, vy, vz;
  if (gq_build_dev_model(desc, &M, &vx, &vy, &vz, err, (size_t)errlen)) return -1;
  static std::vector<float> hf_heights;
  gq_hfield_heights(desc, &hf_heights);
  M.hf_data = hf_heights.empty() ? nullptr : hf_heights.data();
  gq::ResetArgs a{};
  a.model = &M; a.vx = vx.data(); a.vy = vy.data(); a.vz = vz.data(); a.mask = mask; a.qpos_new = qpos_new; a.qvel_new = qvel_new;
  a.qpos = qpos; a.qvel = qvel; a.qacc = qacc; a.warm = warm; a.applied = applied; a.time = time; a.cmd = cmd;
  a.friction_next = friction_next; a.step_num = step_num; a.episode = episode; a.lift_failed = lift_failed;
  a.lift_pending = lift_pending;
  emu_fill_cfg(&a.cfg, cfg);
  for (int e = 0; e < n_envs; e++) {
    if (mask && !mask[e]) continue;
    emu_run_wave((unsigned)e, (unsigned)n_envs, [&]() {
      __shared__ gq::WaveMem W;
      if ((M.nbox > 0 || M.hf_nrow > 0)) gq::reset_wave<true>(a, W); else gq::reset_wave<false>(a, W);
    });
  }
  return 0;
}

/* the kernel's pair routines (csrc/gq_pairs.h) called directly, for a side-by-side with the oracle's restatement
 * (tests/test_kernel_emulated.py::test_pair_routines_kernel_equals_oracle): out = n x [dist, pos[3], nrm[3]] */
extern "C" int emu_capsule_box(const float* p0, const float* p1, float r, const float* bc, const float* bR, const float* bh, float margin, float* out) {
  gq::PairHit H;
  gq::capsule_box(gq::v3(p0[0], p0[1], p0[2]), gq::v3(p1[0], p1[1], p1[2]), r, gq::v3(bc[0], bc[1], bc[2]), bR, gq::v3(bh[0], bh[1], bh[2]), margin, H);
  for (int q = 0; q < H.n; q++) { out[7 * q] = H.dist[q]; out[7 * q + 1] = H.pos[q].x; out[7 * q + 2] = H.pos[q].y; out[7 * q + 3] = H.pos[q].z; out[7 * q + 4] = gq::hit_nrm(H, q).x; out[7 * q + 5] = gq::hit_nrm(H, q).y; out[7 * q + 6] = gq::hit_nrm(H, q).z; }
  return H.n;
}
extern "C" int emu_box_box(const float* ca, const float* Ra, const float* ha, const float* cb, const float* Rb, const float* hb, float margin, float* out) {
  gq::PairHit H;
  gq::box_box(gq::v3(ca[0], ca[1], ca[2]), Ra, gq::v3(ha[0], ha[1], ha[2]), gq::v3(cb[0], cb[1], cb[2]), Rb, gq::v3(hb[0], hb[1], hb[2]), margin, H);
  for (int q = 0; q < H.n; q++) { out[7 * q] = H.dist[q]; out[7 * q + 1] = H.pos[q].x; out[7 * q + 2] = H.pos[q].y; out[7 * q + 3] = H.pos[q].z; out[7 * q + 4] = gq::hit_nrm(H, q).x; out[7 * q + 5] = gq::hit_nrm(H, q).y; out[7 * q + 6] = gq::hit_nrm(H, q).z; }
  return H.n;
}
/* the kernel's convex routine (csrc/gq_convex.h: GJK + EPA, one wavefront per pair) called directly on two shapes - vertex clouds (h = NULL)
 * or analytic boxes (V = NULL) - for a side-by-side with the oracle's restatement: out = dist, pos[3], nrm[3] */
extern "C" int emu_convex(const float* VA, int na, const float* hA, const float* RA, const float* tA, float rA,
                          const float* VB, int nb, const float* hB, const float* RB, const float* tB, float rB, float margin, float* out) {
  std::vector<float> vx, vy, vz;
  for (int i = 0; i < na; i++) { vx.push_back(VA[3 * i]); vy.push_back(VA[3 * i + 1]); vz.push_back(VA[3 * i + 2]); }
  for (int i = 0; i < nb; i++) { vx.push_back(VB[3 * i]); vy.push_back(VB[3 * i + 1]); vz.push_back(VB[3 * i + 2]); }
  if (vx.empty()) { vx.push_back(0); vy.push_back(0); vz.push_back(0); }
  static float shp[GQ_CVX_SHP_WORDS], poly[GQ_CVX_POLY_WORDS];
  int hit = 0;
  emu_run_wave(0, 1, [&]() {
    gq::CvxShape A, B;
    A.kind = VA ? 0 : 1; A.adr = 0; A.num = na; A.pm = -1; A.r = rA; A.t = gq::v3(tA[0], tA[1], tA[2]); A.h = hA ? gq::v3(hA[0], hA[1], hA[2]) : gq::v3(0, 0, 0);
    B.kind = VB ? 0 : 1; B.adr = na; B.num = nb; B.pm = -1; B.r = rB; B.t = gq::v3(tB[0], tB[1], tB[2]); B.h = hB ? gq::v3(hB[0], hB[1], hB[2]) : gq::v3(0, 0, 0);
    for (int i = 0; i < 9; i++) { A.R[i] = RA[i]; B.R[i] = RB[i]; }
    gq::cvx_shape_store(shp, A); gq::cvx_shape_store(shp + GQ_CVX_SHAPE_WORDS, B);
    gq::wave_barrier();
    const bool h = gq::cvx_pair_wave(shp, poly, vx.data(), vy.data(), vz.data(), margin);
    if (gq::lane_id() == 0) hit = h ? 1 : 0;
  });
  if (hit) { const float* o = shp + 2 * GQ_CVX_SHAPE_WORDS; out[0] = o[0]; out[1] = o[4]; out[2] = o[5]; out[3] = o[6]; out[4] = o[1]; out[5] = o[2]; out[6] = o[3]; }
  return hit;
}
