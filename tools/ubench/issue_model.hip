// Issue model of ONE wavefront alone on its SIMD (gfx950 / MI355X) - what a launch's tail waves live by.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/issue_model tools/ubench/issue_model.hip && /tmp/issue_model
// Every test times N copies of an instruction pattern between two s_memtime reads (shader clock) in a single 64-lane workgroup
// and prints cycles per instruction.  Patterns: dependent / independent v_fma_f32 chains (inline asm: no packing, no reordering),
// SALU chains, VALU + SALU interleaved, LDS read latency (dependent) and throughput (independent), wave-uniform LDS reads,
// v_readlane -> SALU use, exec-mask branch blocks (taken / skipped), scalar-cache hit, vector L1/L2 hit.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
#define REP256(x) REP64(x) REP64(x) REP64(x) REP64(x)

__device__ __forceinline__ long long now() { long long t; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t)); return t; }

__global__ void __launch_bounds__(64) k_issue(float* out, long long* cyc, const float* gmem, const int* idx) {
  __shared__ float lds[4096];
  const int lane = threadIdx.x;
  for (int i = lane; i < 4096; i += 64) lds[i] = (float)((i * 7 + 1) & 1023) * 4.0f; // holds byte offsets for pointer chasing
  __syncthreads();
  float a = 1.0001f, b = 0.5f, x0 = lane, x1 = lane + 1, x2 = lane + 2, x3 = lane + 3, x4 = lane + 4, x5 = lane + 5, x6 = lane + 6, x7 = lane + 7;
  int s0 = 1, s1 = 2, s2 = 3, s3 = 4;
  long long t0, t1;
  int n = 0;
  // 0: dependent FMA chain
  t0 = now(); asm volatile(REP256("v_fma_f32 %0, %0, %1, %2\n") : "+v"(x0) : "v"(a), "v"(b)); t1 = now(); cyc[n++] = t1 - t0;
  // 1: 2 independent chains
  t0 = now(); asm volatile(REP256("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n") : "+v"(x0), "+v"(x1) : "v"(a), "v"(b)); t1 = now(); cyc[n++] = t1 - t0;
  // 2: 4 independent chains
  t0 = now(); asm volatile(REP256("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b)); t1 = now(); cyc[n++] = t1 - t0;
  // 3: 8 independent chains
  t0 = now(); asm volatile(REP64("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b)); t1 = now(); cyc[n++] = t1 - t0;
  // 4: dependent SALU chain
  t0 = now(); asm volatile(REP256("s_add_i32 %0, %0, %1\n") : "+s"(s0) : "s"(s1) : "scc"); t1 = now(); cyc[n++] = t1 - t0;
  // 5: 4 independent SALU
  t0 = now(); asm volatile(REP256("s_add_i32 %0, %0, 1\n s_add_i32 %1, %1, 1\n s_add_i32 %2, %2, 1\n s_add_i32 %3, %3, 1\n") : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc"); t1 = now(); cyc[n++] = t1 - t0;
  // 6: dependent VALU interleaved with independent SALU (pairs)
  t0 = now(); asm volatile(REP256("v_fma_f32 %0, %0, %2, %3\n s_add_i32 %1, %1, 1\n") : "+v"(x0), "+s"(s0) : "v"(a), "v"(b) : "scc"); t1 = now(); cyc[n++] = t1 - t0;
  // 7: dependent LDS pointer chase (ds_read_b32 -> address of the next)
  { int p = lane * 4; t0 = now(); asm volatile(REP64("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n v_cvt_u32_f32 %0, %0\n") : "+v"(p)); t1 = now(); cyc[n++] = t1 - t0; x1 += p; }
  // 8: independent LDS reads, 8 in flight, one wait per 8
  { int p = lane * 4; float r0, r1, r2, r3, r4, r5, r6, r7;
    t0 = now(); asm volatile(REP64("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:256\n ds_read_b32 %2, %8 offset:512\n ds_read_b32 %3, %8 offset:768\n ds_read_b32 %4, %8 offset:1024\n ds_read_b32 %5, %8 offset:1280\n ds_read_b32 %6, %8 offset:1536\n ds_read_b32 %7, %8 offset:1792\n s_waitcnt lgkmcnt(0)\n")
      : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(p)); t1 = now(); cyc[n++] = t1 - t0; x2 += r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7; }
  // 9: wave-uniform LDS read + dependent use (broadcast read, wait, fma)
  { int p = 0; float r; t0 = now(); asm volatile(REP64("ds_read_b32 %1, %2\n s_waitcnt lgkmcnt(0)\n v_fma_f32 %0, %0, %1, %1\n") : "+v"(x3), "=&v"(r) : "v"(p)); t1 = now(); cyc[n++] = t1 - t0; }
  // 10: v_readlane -> SALU use -> v_mov back (the bcast idiom), dependent
  t0 = now(); asm volatile(REP64("v_readlane_b32 %1, %0, 3\n s_add_i32 %1, %1, 1\n v_mov_b32 %0, %1\n") : "+v"(x4), "+s"(s0) : : "scc"); t1 = now(); cyc[n++] = t1 - t0;
  // 11: exec-mask if-block, body executed by lanes < 18 (not skipped)
  { int lim = 18; t0 = now(); asm volatile(REP64("v_cmp_gt_i32 vcc, %2, %1\n s_and_saveexec_b64 s[40:41], vcc\n s_cbranch_execz 1\n v_fma_f32 %0, %0, %3, %4\n s_or_b64 exec, exec, s[40:41]\n") : "+v"(x5) : "v"(lane), "v"(lim), "v"(a), "v"(b) : "vcc", "s40", "s41"); t1 = now(); cyc[n++] = t1 - t0; }
  // 12: exec-mask if-block whose branch IS taken (no lane active): skip
  { int lim = -1; t0 = now(); asm volatile(REP64("v_cmp_gt_i32 vcc, %2, %1\n s_and_saveexec_b64 s[40:41], vcc\n s_cbranch_execz 1\n v_fma_f32 %0, %0, %3, %4\n s_or_b64 exec, exec, s[40:41]\n") : "+v"(x6) : "v"(lane), "v"(lim), "v"(a), "v"(b) : "vcc", "s40", "s41"); t1 = now(); cyc[n++] = t1 - t0; }
  // 13: scalar load, cache hit, dependent (address from the previous result is not needed: same address, wait each)
  { unsigned long long base = (unsigned long long)idx; int r; t0 = now(); asm volatile(REP64("s_load_dword %0, %1, 0x0\n s_waitcnt lgkmcnt(0)\n") : "=s"(r) : "s"(base)); t1 = now(); cyc[n++] = t1 - t0; s1 += r; }
  // 14: vector global load, same line every time (L1 hit), wait each
  { const float* q = gmem + lane; float r; t0 = now(); asm volatile(REP64("global_load_dword %0, %1, off\n s_waitcnt vmcnt(0)\n") : "=v"(r) : "v"(q)); t1 = now(); cyc[n++] = t1 - t0; x7 += r; }
  // 15: vector global load with glc (misses L1, L2 hit), wait each
  { const float* q = gmem + lane; float r; t0 = now(); asm volatile(REP64("global_load_dword %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)\n") : "=v"(r) : "v"(q)); t1 = now(); cyc[n++] = t1 - t0; x7 += r; }
  // 16: DPP row_shr add chain (the wave_sum ladder step), dependent
  t0 = now(); asm volatile(REP256("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n") : "+v"(x0)); t1 = now(); cyc[n++] = t1 - t0;
  // 17: v_rcp_f32 dependent chain (transcendental)
  t0 = now(); asm volatile(REP64("v_rcp_f32 %0, %0\n") : "+v"(x1)); t1 = now(); cyc[n++] = t1 - t0;
  // 18: ds_write_b32 stream (fire and forget), then one wait
  { int p = lane * 4; t0 = now(); asm volatile(REP64("ds_write_b32 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : : "v"(p), "v"(x2)); t1 = now(); cyc[n++] = t1 - t0; }
  // 19: ds_bpermute dependent chain
  { int p = ((lane + 1) & 63) * 4; t0 = now(); asm volatile(REP64("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)\n") : "+v"(x3) : "v"(p)); t1 = now(); cyc[n++] = t1 - t0; }
  // 20: ds_write_b32, lanes >= 18 all write lane 17's address (the clamped-lane idiom), stream + one wait
  { int p = (lane < 18 ? lane : 17) * 4; t0 = now(); asm volatile(REP64("ds_write_b32 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : : "v"(p), "v"(x2)); t1 = now(); cyc[n++] = t1 - t0; }
  // 21: ds_write_b32, every lane the same address
  { int p = 0; t0 = now(); asm volatile(REP64("ds_write_b32 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : : "v"(p), "v"(x2)); t1 = now(); cyc[n++] = t1 - t0; }
  // 22: scalar branch, not taken
  t0 = now(); asm volatile(REP64("s_cmp_eq_u32 %0, -7\n s_cbranch_scc1 1\n s_add_i32 %0, %0, 0\n") : "+s"(s0) : : "scc"); t1 = now(); cyc[n++] = t1 - t0;
  // 23: scalar branch, taken (skips one instruction)
  t0 = now(); asm volatile(REP64("s_cmp_lg_u32 %0, -7\n s_cbranch_scc1 1\n s_add_i32 %0, %0, 0\n") : "+s"(s0) : : "scc"); t1 = now(); cyc[n++] = t1 - t0;
  // 24: v_cndmask select instead of a block: v_cmp + v_cndmask + fma
  { int lim = 18; t0 = now(); asm volatile(REP64("v_cmp_gt_i32 vcc, %2, %1\n v_fma_f32 %5, %0, %3, %4\n v_cndmask_b32 %0, %0, %5, vcc\n") : "+v"(x5) : "v"(lane), "v"(lim), "v"(a), "v"(b), "v"(x6) : "vcc"); t1 = now(); cyc[n++] = t1 - t0; }
  // 25: global_store_dword, all lanes one address, stream of 16 + wait
  { float* q = out + 64; t0 = now(); asm volatile(REP8("global_store_dword %0, %1, off\n global_store_dword %0, %1, off\n") "s_waitcnt vmcnt(0)\n" : : "v"(q), "v"(x2) : "memory"); t1 = now(); cyc[n++] = t1 - t0; }
  // 26: global_store_dword, lane-consecutive addresses, stream of 16 + wait
  { float* q = out + 64 + lane; t0 = now(); asm volatile(REP8("global_store_dword %0, %1, off\n global_store_dword %0, %1, off\n") "s_waitcnt vmcnt(0)\n" : : "v"(q), "v"(x2) : "memory"); t1 = now(); cyc[n++] = t1 - t0; }
  out[lane] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + (float)(s0 + s1 + s2 + s3);
}

int main() {
  float* out; long long* cyc; float* gmem; int* idx;
  hipMalloc(&out, 4096); hipMalloc(&cyc, 64 * 8); hipMalloc(&gmem, 4096); hipMalloc(&idx, 256);
  hipMemset(gmem, 0, 4096); hipMemset(idx, 0, 256);
  long long h[32];
  const char* name[] = {"v_fma dependent chain", "v_fma 2 independent chains", "v_fma 4 independent chains", "v_fma 8 independent chains", "s_add dependent chain",
                        "s_add 4 independent", "v_fma dep + s_add pair", "ds_read_b32 pointer chase (+cvt)", "ds_read_b32 x8 in flight + wait", "uniform ds_read + wait + fma",
                        "readlane -> s_add -> v_mov", "if (lane<18) { fma } block", "if (none) { } skipped block", "s_load hit + wait", "global_load L1 hit + wait", "global_load sc0 sc1 (L2) + wait",
                        "v_add_f32_dpp row_shr chain", "v_rcp_f32 dependent chain", "ds_write_b32 x64 + wait", "ds_bpermute dependent chain",
                        "ds_write clamped lanes x64 + wait", "ds_write same address x64 + wait", "s_cmp + s_cbranch not taken + s_add", "s_cmp + s_cbranch taken", "v_cmp + fma + v_cndmask (select)", "global_store same addr x16 + wait", "global_store consecutive x16 + wait"};
  const int per[] = {256, 512, 1024, 512, 256, 1024, 512, 64 * 3, 64 * 9, 64 * 3, 64 * 3, 64 * 5, 64 * 5, 64 * 2, 64 * 2, 64 * 2, 256, 64, 65, 64 * 2, 65, 65, 64 * 3, 64 * 2, 64 * 3, 17, 17};
  const int reps[] = {256, 256, 256, 64, 256, 256, 256, 64, 64, 64, 64, 64, 64, 64, 64, 64, 256, 64, 1, 64, 1, 1, 64, 64, 64, 1, 1};
  for (int it = 0; it < 3; it++) { hipLaunchKernelGGL(k_issue, dim3(1), dim3(64), 0, 0, out, cyc, gmem, idx); hipDeviceSynchronize(); }
  hipMemcpy(h, cyc, sizeof(long long) * 27, hipMemcpyDeviceToHost);
  printf("one wavefront alone on its SIMD (s_memtime shader cycles; timer overhead ~%d cycles not subtracted)\n", 40);
  for (int i = 0; i < 27; i++) printf("  %-36s %8lld cycles  = %6.1f per pattern repeat, %5.2f per instruction\n", name[i], h[i], (double)h[i] / reps[i], (double)h[i] / per[i]);
  return 0;
}
