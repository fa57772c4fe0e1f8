// VALU issue-rate microbenchmark for gfx950 (MI355X): how many cycles does a wave64 v_fma_f32 take
//   (a) in one DEPENDENT chain,  (b) in 8 INDEPENDENT chains,
// with 1, 2, 4 and 8 resident waves per SIMD?   Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/fma_issue tools/ubench/fma_issue.hip && /tmp/fma_issue
// Every wave times itself with s_memtime (shader clock); the table prints cycles per FMA as seen by ONE wave and the
// resulting FMA issue interval of the SIMD (= per-wave cycles / resident waves).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define N_FMA 4096
template <int INDEP>
__global__ void __launch_bounds__(64) fma_kernel(float* out, long long* cyc, float a, float b) {
  float x[8];
#pragma unroll
  for (int k = 0; k < 8; k++) x[k] = (float)threadIdx.x + (float)k;
  const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < N_FMA / 64; i++) {
    if (INDEP) {
#pragma unroll
      for (int u = 0; u < 8; u++) {
#pragma unroll
        for (int k = 0; k < 8; k++) x[k] = __builtin_fmaf(x[k], a, b);
      }
    } else {
#pragma unroll
      for (int u = 0; u < 64; u++) x[0] = __builtin_fmaf(x[0], a, b);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.0f;
#pragma unroll
  for (int k = 0; k < 8; k++) s += x[k];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int simds = p.multiProcessorCount * 4;
  printf("%s: %d CUs, %d SIMDs; %d FMAs per wave\n", p.name, p.multiProcessorCount, simds, N_FMA);
  printf("%-12s %-6s %14s %16s %12s\n", "chains", "waves", "cyc/FMA/wave", "SIMD cyc/FMA", "wall us");
  for (int indep = 0; indep < 2; indep++)
    for (int w : {1, 2, 4, 8}) {
      const int grid = simds * w;
      float* out; long long* cyc;
      hipMalloc(&out, grid * 64 * sizeof(float)); hipMalloc(&cyc, grid * sizeof(long long));
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        if (indep) hipLaunchKernelGGL(fma_kernel<1>, dim3(grid), dim3(64), 0, 0, out, cyc, 1.0001f, 0.5f);
        else hipLaunchKernelGGL(fma_kernel<0>, dim3(grid), dim3(64), 0, 0, out, cyc, 1.0001f, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      std::vector<long long> h(grid);
      hipMemcpy(h.data(), cyc, grid * sizeof(long long), hipMemcpyDeviceToHost);
      std::sort(h.begin(), h.end());
      const double med = (double)h[grid / 2] / N_FMA;
      printf("%-12s %-6d %14.2f %16.2f %12.1f\n", indep ? "8 indep" : "1 dependent", w, med, med / w, ms * 1e3);
      hipFree(out); hipFree(cyc);
    }
  return 0;
}
