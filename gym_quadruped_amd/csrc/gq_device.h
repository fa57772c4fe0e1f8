/*
 * gq_device.h - wave-level primitives for the gfx950 kernels (one env per 64-lane wavefront, one wavefront
 * per workgroup).  Cross-lane primitives must be called from wave-uniform control flow.
 */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GQ_WAVE 64

/* Pointers that the kernels load FROM MEMORY (the device-resident argument block) have no address space the compiler
 * could infer: every access through them becomes a FLAT instruction (64-bit VGPR addresses, both wait counters).  They
 * all point to global memory, so they are cast into address space 1 where they are used: global_load / global_store
 * with a scalar base again. */
#define GQ_GLOBAL __attribute__((address_space(1)))
/* The model (GqDevModel and the collision clouds) is never written while a kernel runs: it is read through the constant
 * address space, so that its loads are invariant - scalar loads where the address is uniform, free to be hoisted above
 * stores and to be merged where the same word is read twice. */
#define GQ_MODEL __attribute__((address_space(4)))
/* LDS (the per-wave working set): for pointers that cross a function boundary (gq_convex.h) */
#define GQ_LDS __attribute__((address_space(3)))

namespace gq {

/* The lane index is deliberately opaque to the optimiser (empty asm volatile): per-lane address arithmetic then stays
 * next to its use instead of being hoisted to the kernel prologue and kept live (or spilled) across the whole step. */
#ifndef GQ_WPB
#define GQ_WPB 1 /* wavefronts (= envs) per workgroup */
#endif
__device__ __forceinline__ int lane_id() {
  int l = GQ_WPB == 1 ? (int)threadIdx.x : (int)(threadIdx.x & (GQ_WAVE - 1));
  asm volatile("" : "+v"(l));
  return l;
}
/* index of this wavefront's env within the launch */
__device__ __forceinline__ int wave_index() {
  return GQ_WPB == 1 ? (int)blockIdx.x : (int)blockIdx.x * GQ_WPB + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
}

/* LDS hand-off between lanes of the single wavefront of this workgroup.  With a 64-thread workgroup the
 * s_barrier degenerates (LLVM drops it for single-wave groups) and what remains is the lgkmcnt wait + the
 * compiler-level ordering of LDS accesses. */
#ifndef GQ_FENCE_BARRIER
#define GQ_FENCE_BARRIER 1
#endif
__device__ __forceinline__ void wave_barrier() {
#if GQ_FENCE_BARRIER
  /* the LDS executes the DS instructions of one wavefront in order: a ds_read issued after a ds_write of another lane of the
   * same wave sees the data without any wait - only the COMPILER must not reorder them */
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
#else
  __syncthreads();
#endif
}

/* broadcast lane `src` (wave-uniform index) - v_readlane_b32 */
__device__ __forceinline__ float bcast(float v, int src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
}
__device__ __forceinline__ int bcast(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
template <int SRC>
__device__ __forceinline__ float readlane(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), SRC));
}

/* butterfly exchange: DPP within rows of 16 / quad perms, ds_swizzle / permlane for the wider strides */
__device__ __forceinline__ float shfl_xor(float v, int m) { return __shfl_xor(v, m, GQ_WAVE); }
__device__ __forceinline__ int shfl_xor(int v, int m) { return __shfl_xor(v, m, GQ_WAVE); }

/* a value the program knows to be wave-uniform but the compiler does not (e.g. read from LDS): v_readfirstlane */
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float uniformf(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

/* read lane `src` (per-lane index) - ds_bpermute_b32 */
__device__ __forceinline__ float shfl_idx(float v, int src) { return __shfl(v, src, GQ_WAVE); }
__device__ __forceinline__ int shfl_idx(int v, int src) { return __shfl(v, src, GQ_WAVE); }

__device__ __forceinline__ uint64_t ballot(bool p) { return __ballot(p); }
__device__ __forceinline__ int popc64(uint64_t m) { return __popcll(m); }
__device__ __forceinline__ int ffs64(uint64_t m) { return __ffsll((long long)m) - 1; } /* index of the lowest set bit */

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

/* full-wave reductions, result in every lane.  row_shr / row_bcast DPP ladder + readlane(63). */
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_mov<0x111>(v);  /* row_shr:1 */
  v += dpp_mov<0x112>(v);  /* row_shr:2 */
  v += dpp_mov<0x114>(v);  /* row_shr:4 */
  v += dpp_mov<0x118>(v);  /* row_shr:8  -> lane 15 of each row holds the row sum */
  /* combine the four row sums held in lanes 15, 31, 47, 63 */
  float r0 = readlane<15>(v), r1 = readlane<31>(v), r2 = readlane<47>(v), r3 = readlane<63>(v);
  return (r0 + r1) + (r2 + r3);
}
/* min / max: the same row_shr ladder; out-of-row lanes read their own value (bound_ctrl off, old = self) */
template <int CTRL>
__device__ __forceinline__ float dpp_mov_self(float v) {
  const int iv = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(iv, iv, CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_min(float v) {
  v = fminf(v, dpp_mov_self<0x111>(v)); v = fminf(v, dpp_mov_self<0x112>(v));
  v = fminf(v, dpp_mov_self<0x114>(v)); v = fminf(v, dpp_mov_self<0x118>(v));
  return fminf(fminf(readlane<15>(v), readlane<31>(v)), fminf(readlane<47>(v), readlane<63>(v)));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_mov_self<0x111>(v)); v = fmaxf(v, dpp_mov_self<0x112>(v));
  v = fmaxf(v, dpp_mov_self<0x114>(v)); v = fmaxf(v, dpp_mov_self<0x118>(v));
  return fmaxf(fmaxf(readlane<15>(v), readlane<31>(v)), fmaxf(readlane<47>(v), readlane<63>(v)));
}

/* inclusive prefix sum over the lanes (lane l gets v_0 + .. + v_l): row_shr ladder with zero fill inside the rows of 16, then
 * the totals of the lower rows (lanes 15 / 31 / 47, v_readlane) are added to the upper ones */
template <int CTRL>
__device__ __forceinline__ int dpp_mov_zero(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
__device__ __forceinline__ int wave_incl_scan(int v) {
  v += dpp_mov_zero<0x111>(v); v += dpp_mov_zero<0x112>(v); v += dpp_mov_zero<0x114>(v); v += dpp_mov_zero<0x118>(v);
  const int r0 = __builtin_amdgcn_readlane(v, 15), r1 = __builtin_amdgcn_readlane(v, 31), r2 = __builtin_amdgcn_readlane(v, 47);
  const int row = ((int)threadIdx.x & (GQ_WAVE - 1)) >> 4;
  return v + (row >= 1 ? r0 : 0) + (row >= 2 ? r1 : 0) + (row >= 3 ? r2 : 0);
}

/* sum over the four lanes of the lane's quad, result in all four (two quad_perm DPP butterflies: [1,0,3,2], [2,3,0,1]) */
__device__ __forceinline__ float quad_sum(float v) {
  v += dpp_mov<0xB1>(v); /* every source lane of a quad_perm is inside the quad: the plain form folds into v_add_f32_dpp */
  v += dpp_mov<0x4E>(v);
  return v;
}

/* optimisation barrier: the value becomes opaque to the compiler (no code is emitted) */
__device__ __forceinline__ void opaque(int& v) { asm volatile("" : "+v"(v)); }
template <class T> __device__ __forceinline__ GQ_GLOBAL T* gptr(T* p) { return (GQ_GLOBAL T*)p; }
template <class T> __device__ __forceinline__ const GQ_MODEL T* mptr(const T* p) { return (const GQ_MODEL T*)p; }
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
/* issue priority of this wave among the waves of its SIMD (0 lowest .. 3); p is wave-uniform */
__device__ __forceinline__ void wave_priority(int p) {
  p = __builtin_amdgcn_readfirstlane(p);
  if (p <= 0) __builtin_amdgcn_s_setprio(0);
  else if (p == 1) __builtin_amdgcn_s_setprio(1);
  else if (p == 2) __builtin_amdgcn_s_setprio(2);
  else __builtin_amdgcn_s_setprio(3);
}
__device__ __forceinline__ int opaque_lane(int l) { asm volatile("" : "+v"(l)); return l; }
template <class T> __device__ __forceinline__ const T* opaque_ptr(const T* p) { asm volatile("" : "+v"(p)); return p; } /* per-lane pointer */
__device__ __forceinline__ void opaque_s(int& v) { asm volatile("" : "+s"(v)); } /* wave-uniform value */
/* pin(): a wave-uniform value the program has just loaded (a pointer out of the device-resident argument block, a model scalar) is
 * tied to its SGPRs here.  Two things follow, and both are the point.  (1) Every scalar load written in front of a group of pins is
 * ISSUED before the first pin and all of them are waited for ONCE (the pins are volatile: loads do not sink below them) - instead of one
 * s_load + s_waitcnt lgkmcnt(0) at each use, which is what the register allocator makes of an invariant load whose value it would have to
 * keep (it re-materialises the load in front of every use: m.timestep was fetched five times between the Euler step and the observation
 * row, each time with its own exposed scalar-cache round trip).  (2) From here on the value is the asm's result, which cannot be
 * re-materialised: it stays in its SGPRs (or a v_writelane slot) until its last use. */
template <class T> __device__ __forceinline__ void pin(T*& p) { asm volatile("" : "+s"(p)); }
__device__ __forceinline__ void pin(int& v) { asm volatile("" : "+s"(v)); }
__device__ __forceinline__ void pin(float& v) { asm volatile("" : "+s"(v)); }
__device__ __forceinline__ void pin(double& v) { asm volatile("" : "+s"(v)); }

/* ---- mailbox traffic of the closed-loop persistent rollout (gq_kernels.hip mailbox_step_kernel): words that another wavefront -
 * possibly on another XCD, behind another L2 - reads or writes while this kernel runs.  Agent-scope relaxed atomics: the
 * loads / stores carry sc1 (served by / written through to the device-coherent level), the read-modify-writes are performed
 * there.  Ordering is explicit: publish_fence() before the store of a sequence word (every earlier store of this wave has been
 * acknowledged), the consumer reads data only after it has seen the sequence word. */
__device__ __forceinline__ int ld_pub(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_pub(const float* p) {
  return __builtin_bit_cast(float, __hip_atomic_load(reinterpret_cast<const uint32_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_pub(int32_t* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_pub(float* p, float v) {
  __hip_atomic_store(reinterpret_cast<uint32_t*>(p), __builtin_bit_cast(uint32_t, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
/* a load of per-env state that another wavefront may have written earlier in THIS launch (closed-loop rollout: an env changes hands between
 * wavefronts, i.e. between CUs): device scope makes it miss the CU's vector L1, which holds whatever the CU read when it last handled the env
 * or one of the envs sharing the line - gfx950 has no instruction that drops the L1 alone when a workgroup sits on one CU (buffer_inv sc0
 * is a no-op there; buffer_inv sc1 also empties the XCD's L2: -14 %).  PUB false: a plain global load. */
template <bool PUB, class T> __device__ __forceinline__ T ldv(const T* p) {
  if constexpr (PUB) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else return *(const GQ_GLOBAL T*)p;
}
__device__ __forceinline__ int add_pub(int32_t* p, int v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool cas_pub(int32_t* p, int expect, int v) { return __hip_atomic_compare_exchange_strong(p, &expect, v, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void publish_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); /* compiler: no store sinks below */
  __builtin_amdgcn_s_waitcnt(0x0F70);                    /* vmcnt(0): every store of this wave has reached its L2 / memory */
}
/* an env of the closed-loop rollout changes hands between wavefronts of ONE XCD (per-XCD ready queues): its rows are coherent in
 * that XCD's L2; what a new owner must not use is its CU's vector L1 (write-through, not snooped) and the scalar cache */
__device__ __forceinline__ void adopt_fence() {
  /* the vector L1 is NOT dropped here (see ldv): every load of the adopted env's mutable rows is a device-scope load.  Scalar cache: no
   * per-env word is read through it today (the compiler only scalarises loads nothing in the kernel can clobber), invalidated anyway */
  __builtin_amdgcn_s_dcache_inv();
  __builtin_amdgcn_s_waitcnt(0xC07F); /* lgkmcnt(0): the invalidate has completed */
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ int xcc_id() { return (int)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xF); } /* HW_REG_XCC_ID */
__device__ __forceinline__ void nap() { __builtin_amdgcn_s_sleep(8); }

/* shader clock (s_memtime) for the stage timers of the debug record */
__device__ __forceinline__ long long cycles() { return (long long)__builtin_readcyclecounter(); }

/* instruction-scheduling fence: nothing moves across it */
__device__ __forceinline__ void sched_fence() { __builtin_amdgcn_sched_barrier(0); }

__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
/* a / b as v_rcp_f32 + v_mul_f32 (2 instructions, <= 2 ulp).  The compiler's own fp32 '/' - even with
 * -fno-hip-fp32-correctly-rounded-divide-sqrt - wraps the reciprocal in v_frexp_mant / v_frexp_exp / v_ldexp range scaling, 8
 * instructions; the operands on the hot paths (residuals, impedances, squared lengths guarded by epsilons) are nowhere near
 * the ends of the fp32 range. */
/* v_sqrt_f32 (1 ulp) without libm's scaling of tiny arguments (sqrtf: v_cmp + 2 v_ldexp + v_cndmask around it) */
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float fdiv(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
/* cos(2 pi x), x in turns (v_cos_f32; absolute error ~1e-6 on [0, 1)) */
__device__ __forceinline__ float fast_cos_turns(float x) { return __builtin_amdgcn_cosf(x); }
/* a^p / b^q for a, b in (0, 1]: exp2(p log2 a - q log2 b) on the transcendental unit (v_log_f32 / v_exp_f32, ~1 ulp each;
 * the library powf is 600 instructions of special-case handling that these arguments never reach) */
__device__ __forceinline__ float fast_pow_ratio(float a, float p, float b, float q) {
  return __builtin_amdgcn_exp2f(p * __builtin_amdgcn_logf(a) - q * __builtin_amdgcn_logf(b));
}
__device__ __forceinline__ float med3(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }

}  // namespace gq
