/*
 * gq_kernels.hip - gfx950 kernels of libgq: one environment per 64-lane wavefront, one wavefront per workgroup
 * (grid = n_envs).  With 4096 envs and <= 128 VGPRs / <= 10 KB LDS per wave the whole batch is resident at once
 * (4 waves per SIMD, 16 per CU); workgroup b lands on XCD b % 8, so consecutive envs spread over all eight L2s and
 * every XCD keeps its own copy of the read-only model block.
 */
#include <gq_device.h>
#include "gq_step_body.h"

namespace gq {

/* step (+ in-kernel auto-reset): a terminated env is re-spawned by the same wavefront - reset_wave, then the reset's
 * own mj_step as a second pass through step_wave - so auto-reset costs no extra launches and only the few
 * terminated envs pay for the second pass. */
template <int SOLVER>
__global__ void __launch_bounds__(GQ_WAVE, 4) step_kernel(FusedArgs a) {
  if (a.s.mask && !a.s.mask[blockIdx.x]) return; /* wave-uniform */
  __shared__ WaveMem W;
  int pass = a.first_pass;
  for (;;) {
    const int term = step_wave<SOLVER>(a.s, W, pass);
    if (pass == 1 || !a.auto_reset || !term) break;
    reset_wave(a.r, W);
    pass = 1;
  }
}

__global__ void __launch_bounds__(GQ_WAVE) reset_kernel(ResetArgs a) {
  if (a.mask && !a.mask[blockIdx.x]) return;
  __shared__ WaveMem W;
  reset_wave(a, W);
}

}  // namespace gq

extern "C" void gq_launch_step(const gq::FusedArgs* a, int n_envs, int solver, hipStream_t stream) {
  if (solver == 1) hipLaunchKernelGGL(gq::step_kernel<1>, dim3(n_envs), dim3(GQ_WAVE), 0, stream, *a);
  else hipLaunchKernelGGL(gq::step_kernel<0>, dim3(n_envs), dim3(GQ_WAVE), 0, stream, *a);
}
extern "C" void gq_launch_reset(const gq::ResetArgs* a, int n_envs, hipStream_t stream) {
  hipLaunchKernelGGL(gq::reset_kernel, dim3(n_envs), dim3(GQ_WAVE), 0, stream, *a);
}
