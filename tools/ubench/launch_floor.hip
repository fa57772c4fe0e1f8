// Launch-floor attribution for the step kernel's grid on gfx950 (MI355X): what does a launch of 4096 one-wave workgroups cost
// before it does any physics?   hipcc --offload-arch=gfx950 -O3 -o /tmp/launch_floor tools/ubench/launch_floor.hip && /tmp/launch_floor
// Variants (all grid = 4096 x 64 threads, __launch_bounds__(64, 4) like gq::step_kernel):
//   empty          returns at once, no LDS
//   lds10k         10 240 B of static LDS per workgroup (touched by one store), returns
//   lds+args       + the step kernel's argument chain: kernarg -> device-resident pointer block -> pending byte
//   lds+rows       + the prologue loads of one env's state rows (qpos f64[19], qvel/warm/applied f32[18], ctrl[12], cmd[4], 3 scalars) into LDS
//   lds+rows+st    + the epilogue stores (qpos, qvel, qacc, warm rows, 227-float observation row, flags): the 1.3 KB a step writes per env
// For every variant: (a) back-to-back time per launch over 2000 launches on one stream (events at both ends only - what bench.py's
// ms_per_step sees), (b) per-launch time with a HIP event pair around EVERY launch (what tools/stage_cuts.py sees: includes the
// event packets), both in microseconds.  rocprofv3 --kernel-trace --stats over this binary gives (c), the begin -> end time stamps.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

struct Args { double* qpos; float* qvel; float* warm; float* applied; float* ctrl; float* cmd; float* time; float* friction; int* step_num;
              unsigned char* pending; float* qacc; float* obs; unsigned char* term; float* reward; };

__global__ void __launch_bounds__(64, 4) k_empty(const Args* A, int mode) {}

template <int MODE>
__global__ void __launch_bounds__(64, 4) k_floor(const Args* __restrict__ A, int flag) {
  __shared__ float W[2560];
  const int lane = threadIdx.x, env = blockIdx.x;
  if (MODE == 0) { if (flag == 12345) W[lane] = 1.0f; return; }
  bool respawn = false;
  if (MODE >= 1) respawn = A->pending[env] != 0;
  if (MODE == 1) { if (respawn) W[lane] = 2.0f; if (flag == 12345) A->reward[env] = W[lane ^ 1]; return; }
  double q = 0.0; float qv = 0, wm = 0, ap = 0, ct = 0, cm = 0, mu = 0, tm = 0; int sn = 0;
  if (lane < 19) q = A->qpos[(size_t)env * 19 + lane];
  if (lane < 18) { qv = A->qvel[(size_t)env * 18 + lane]; wm = A->warm[(size_t)env * 18 + lane]; ap = A->applied[(size_t)env * 18 + lane]; }
  if (lane < 12) ct = A->ctrl[(size_t)env * 12 + lane];
  if (lane < 4) cm = A->cmd[(size_t)env * 4 + lane];
  if (lane == 0) { mu = A->friction[env]; sn = A->step_num[env]; tm = A->time[env]; }
  W[lane] = (float)q; W[64 + lane] = qv; W[128 + lane] = wm; W[192 + lane] = ap; W[256 + lane] = ct; W[320 + lane] = cm + mu + tm + (float)sn;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const float s = W[lane ^ 1] + W[64 + (lane ^ 3)] + W[128 + (lane ^ 5)] + W[192 + (lane ^ 7)] + W[256 + (lane ^ 9)] + W[320 + (lane ^ 2)] + (respawn ? 1.0f : 0.0f);
  if (MODE == 2) { if (flag == 12345 || s == 1.2345e33f) A->reward[env] = s; return; }
  // epilogue stores
  if (lane < 19) A->qpos[(size_t)env * 19 + lane] = q;
  if (lane < 18) { A->qvel[(size_t)env * 18 + lane] = qv; A->qacc[(size_t)env * 18 + lane] = s; A->warm[(size_t)env * 18 + lane] = wm; }
  for (int i = 0; i < 4; i++) { const int k = lane + 64 * i; if (k < 227) A->obs[(size_t)env * 227 + k] = s + (float)k; }
  if (lane == 0) { A->term[env] = 0; A->pending[env] = 0; A->reward[env] = 0.0f; A->step_num[env] = sn; A->time[env] = tm; }
}

template <class F> static int run(const char* name, F launch) {
  const int N = 2000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 200; i++) launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); for (int i = 0; i < N; i++) launch(); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  const float b2b = ms * 1e3f / N;
  std::vector<hipEvent_t> ev(2 * 300);
  for (auto& e : ev) CK(hipEventCreate(&e));
  for (int i = 0; i < 300; i++) { CK(hipEventRecord(ev[2 * i])); launch(); CK(hipEventRecord(ev[2 * i + 1])); }
  CK(hipDeviceSynchronize());
  std::vector<float> t(300);
  for (int i = 0; i < 300; i++) CK(hipEventElapsedTime(&t[i], ev[2 * i], ev[2 * i + 1]));
  std::sort(t.begin(), t.end());
  printf("%-16s back-to-back %6.2f us/launch   event pair per launch: median %6.2f us  p10 %6.2f\n", name, b2b, t[150] * 1e3f, t[30] * 1e3f);
  return 0;
}

int main() {
  const int N = 4096;
  Args h;
  CK(hipMalloc(&h.qpos, N * 19 * 8)); CK(hipMalloc(&h.qvel, N * 18 * 4)); CK(hipMalloc(&h.warm, N * 18 * 4)); CK(hipMalloc(&h.applied, N * 18 * 4));
  CK(hipMalloc(&h.ctrl, N * 12 * 4)); CK(hipMalloc(&h.cmd, N * 4 * 4)); CK(hipMalloc(&h.time, N * 4)); CK(hipMalloc(&h.friction, N * 4)); CK(hipMalloc(&h.step_num, N * 4));
  CK(hipMalloc(&h.pending, N)); CK(hipMalloc(&h.qacc, N * 18 * 4)); CK(hipMalloc(&h.obs, N * 227 * 4)); CK(hipMalloc(&h.term, N)); CK(hipMalloc(&h.reward, N * 4));
  CK(hipMemset(h.qpos, 0, N * 19 * 8)); CK(hipMemset(h.qvel, 0, N * 18 * 4)); CK(hipMemset(h.warm, 0, N * 18 * 4)); CK(hipMemset(h.applied, 0, N * 18 * 4));
  CK(hipMemset(h.ctrl, 0, N * 12 * 4)); CK(hipMemset(h.cmd, 0, N * 16)); CK(hipMemset(h.time, 0, N * 4)); CK(hipMemset(h.friction, 0, N * 4)); CK(hipMemset(h.step_num, 0, N * 4));
  CK(hipMemset(h.pending, 0, N));
  Args* d; CK(hipMalloc(&d, sizeof(Args))); CK(hipMemcpy(d, &h, sizeof(Args), hipMemcpyHostToDevice));
  for (int n : {4096, 1024, 256}) {
    printf("grid = %d workgroups x 64 threads\n", n);
    if (run("empty", [&] { hipLaunchKernelGGL(k_empty, dim3(n), dim3(64), 0, 0, d, 0); })) return 1;
    if (run("lds10k", [&] { hipLaunchKernelGGL(k_floor<0>, dim3(n), dim3(64), 0, 0, d, 0); })) return 1;
    if (run("lds+args", [&] { hipLaunchKernelGGL(k_floor<1>, dim3(n), dim3(64), 0, 0, d, 0); })) return 1;
    if (run("lds+rows", [&] { hipLaunchKernelGGL(k_floor<2>, dim3(n), dim3(64), 0, 0, d, 0); })) return 1;
    if (run("lds+rows+stores", [&] { hipLaunchKernelGGL(k_floor<3>, dim3(n), dim3(64), 0, 0, d, 0); })) return 1;
  }
  return 0;
}
