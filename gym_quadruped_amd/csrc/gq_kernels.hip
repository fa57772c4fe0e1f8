/*
 * gq_kernels.hip - gfx950 kernels of libgq: one environment per 64-lane wavefront, one wavefront per workgroup
 * (grid = n_envs).  With 4096 envs the launch fills the 256 CUs 16 waves deep; workgroup b lands on XCD b % 8, so
 * consecutive envs spread over all eight L2s and every XCD holds its own copy of the (read-only) model block.
 */
#include <gq_device.h>
#include "gq_step_body.h"

namespace gq {

__global__ void __launch_bounds__(GQ_WAVE) step_kernel(StepArgs a) {
  if (a.mask && !a.mask[blockIdx.x]) return; /* wave-uniform */
  __shared__ WaveMem W;
  __shared__ float acc[4][21];
  step_wave(a, W, acc);
}

__global__ void __launch_bounds__(GQ_WAVE) reset_kernel(ResetArgs a) {
  if (a.mask && !a.mask[blockIdx.x]) return;
  __shared__ WaveMem W;
  reset_wave(a, W);
}

}  // namespace gq

extern "C" void gq_launch_step(const gq::StepArgs* a, int n_envs, hipStream_t stream) {
  hipLaunchKernelGGL(gq::step_kernel, dim3(n_envs), dim3(GQ_WAVE), 0, stream, *a);
}
extern "C" void gq_launch_reset(const gq::ResetArgs* a, int n_envs, hipStream_t stream) {
  hipLaunchKernelGGL(gq::reset_kernel, dim3(n_envs), dim3(GQ_WAVE), 0, stream, *a);
}
