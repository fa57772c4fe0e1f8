// What does a wavefront of the step kernel's grid pay for its FIRST memory accesses of a launch?  gfx950 (MI355X).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/launch_cold tools/ubench/launch_cold.hip && /tmp/launch_cold
// grid = 4096 one-wave workgroups, launched back to back on one stream like gq_step.  Every wave times (s_memtime), in order:
//   t0  kernarg -> pointer block (two dependent scalar loads: the kernel's argument chain)
//   t1  its env's state row (1.7 KB written by the SAME workgroup index - the same XCD - in the previous launch)
//   t2  a per-lane record of a shared read-only table (the model: 64 KB, read by every wave of every launch), first touch
//   t3  the same record again (vector L1 / L2 hit)
//   t4  another line of the table, first touch, issued TOGETHER with a second row read (overlap check)
// and stores the five latencies; the host prints their mean / p95 over the waves of the last launches.
// Variant B: the table lines of t2 are prefetched (one global load per 128-byte line, discarded) in the same batch as the row of t1.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
struct Block { float* rows; const float* table; unsigned* lat; int pad; };
__device__ __forceinline__ unsigned long long now() { return __builtin_readcyclecounter(); }
template <int PREFETCH>
__global__ void __launch_bounds__(64, 4) k(const Block* __restrict__ B, int launch) {
  const int lane = threadIdx.x, env = blockIdx.x;
  const unsigned long long a0 = now();
  float* rows = B->rows; const float* table = B->table; unsigned* lat = B->lat;
  asm volatile("" : "+s"(rows), "+s"(table), "+s"(lat));
  const unsigned long long a1 = now();
  float* row = rows + (size_t)env * 448;
  float r0 = row[lane], r1 = row[64 + lane], r2 = row[128 + lane];
  float pf = 0.0f;
  if (PREFETCH) pf = table[(lane & 31) * 32];           /* 32 lines x 128 B = the first 4 KB of the table */
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(pf));
  const unsigned long long a2 = now();
  float m0 = table[lane * 12], m1 = table[lane * 12 + 4], m2 = table[lane * 12 + 8];   /* 48-byte records: 3 KB */
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(m0), "+v"(m1), "+v"(m2));
  const unsigned long long a3 = now();
  float n0 = table[lane * 12 + 1], n1 = table[lane * 12 + 5];
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(n0), "+v"(n1));
  const unsigned long long a4 = now();
  float p0 = table[4096 + lane * 16], p1 = row[192 + lane];
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(p0), "+v"(p1));
  const unsigned long long a5 = now();
  const float s = r0 + r1 + r2 + pf + m0 + m1 + m2 + n0 + n1 + p0 + p1;
  for (int i = 0; i < 7; i++) row[64 * i + lane] = s * 1e-30f + (float)(launch + i);   /* the step's 1.7 KB of stores */
  if (lane == 0) { unsigned* L = lat + (size_t)env * 8; L[0] = (unsigned)(a1 - a0); L[1] = (unsigned)(a2 - a1); L[2] = (unsigned)(a3 - a2); L[3] = (unsigned)(a4 - a3); L[4] = (unsigned)(a5 - a4); }
}
int main() {
  const int N = 4096;
  float *rows, *table; unsigned* lat; Block hb, *db;
  CK(hipMalloc(&rows, sizeof(float) * 448 * N)); CK(hipMalloc(&table, sizeof(float) * 16384)); CK(hipMalloc(&lat, sizeof(unsigned) * 8 * N)); CK(hipMalloc(&db, sizeof(Block)));
  CK(hipMemset(rows, 0, sizeof(float) * 448 * N)); CK(hipMemset(table, 0, sizeof(float) * 16384));
  hb.rows = rows; hb.table = table; hb.lat = lat; hb.pad = 0; CK(hipMemcpy(db, &hb, sizeof hb, hipMemcpyHostToDevice));
  std::vector<unsigned> h(8 * N);
  for (int variant = 0; variant < 2; variant++) {
    for (int grid : {4096, 256}) {
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      for (int i = 0; i < 200; i++) { if (variant) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(64), 0, 0, db, i); else hipLaunchKernelGGL(k<0>, dim3(grid), dim3(64), 0, 0, db, i); }
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      for (int i = 0; i < 1000; i++) { if (variant) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(64), 0, 0, db, i); else hipLaunchKernelGGL(k<0>, dim3(grid), dim3(64), 0, 0, db, i); }
      CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      CK(hipMemcpy(h.data(), lat, sizeof(unsigned) * 8 * N, hipMemcpyDeviceToHost));
      printf("%s grid %4d: %.2f us/launch; cycles mean / p95:", variant ? "prefetch" : "plain   ", grid, ms);
      const char* nm[5] = {"args", "row", "table 1st", "table again", "table far + row"};
      for (int c = 0; c < 5; c++) {
        std::vector<unsigned> v(grid); for (int e = 0; e < grid; e++) v[e] = h[8 * e + c];
        std::sort(v.begin(), v.end()); double m = 0; for (auto x : v) m += x;
        printf("  %s %.0f / %u", nm[c], m / grid, v[(size_t)(0.95 * grid)]);
      }
      printf("\n");
    }
  }
  return 0;
}
