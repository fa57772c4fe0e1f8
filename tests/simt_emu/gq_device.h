/*
 * TEST INFRASTRUCTURE - host SIMT emulator shim.
 *
 * tests/simt_emu/ compiles the UNMODIFIED kernel sources of gym_quadruped_amd/csrc/ with g++ and runs one
 * 64-lane "wavefront" as 64 ucontext coroutines, so kernel logic can be checked against the oracle in the
 * GPU-less container.  This header shadows csrc/gq_device.h (it is found first on the include path of the
 * emulator build only) and provides the handful of wave primitives the kernels use.  Cross-lane primitives
 * rendezvous all 64 lanes, therefore kernels must call them from wave-uniform control flow only - the same
 * discipline the hardware versions need for full-wave semantics.  The product never builds or loads this.
 */
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <algorithm>

#define GQ_WAVE 64
#define GQ_GLOBAL
#define GQ_MODEL
#define GQ_LDS
#define __global__
#define __device__
#define __host__
#define __shared__ static
#define __forceinline__ inline
#define __restrict__
#define __launch_bounds__(...)

struct EmuDim3 { unsigned x, y, z; };
extern EmuDim3 threadIdx, blockIdx, blockDim, gridDim;
void emu_yield();                 /* switch to the next lane */
extern uint32_t emu_slot[2][GQ_WAVE];
extern int emu_phase[GQ_WAVE];

namespace gq {
static inline int lane_id() { return (int)threadIdx.x; }
static inline int wave_index() { return (int)blockIdx.x; }
static inline void wave_barrier() { emu_yield(); }

/* deposit a 32-bit value, rendezvous, return pointer to the 64 deposited values (valid until next primitive) */
static inline const uint32_t* exchange(uint32_t v) {
  int l = lane_id();
  int ph = emu_phase[l] ^= 1;
  emu_slot[ph][l] = v;
  emu_yield();
  return emu_slot[ph];
}
static inline float bcast(float v, int src) { uint32_t u; memcpy(&u, &v, 4); const uint32_t* s = exchange(u); float r; memcpy(&r, &s[src & 63], 4); return r; }
static inline int bcast(int v, int src) { const uint32_t* s = exchange((uint32_t)v); return (int)s[src & 63]; }
template <int SRC> static inline float readlane(float v) { return bcast(v, SRC); }
static inline float shfl_xor(float v, int m) { uint32_t u; memcpy(&u, &v, 4); const uint32_t* s = exchange(u); float r; memcpy(&r, &s[lane_id() ^ m], 4); return r; }
static inline int shfl_xor(int v, int m) { const uint32_t* s = exchange((uint32_t)v); return (int)s[lane_id() ^ m]; }
static inline float shfl_idx(float v, int src) { uint32_t u; memcpy(&u, &v, 4); const uint32_t* s = exchange(u); float r; memcpy(&r, &s[src & 63], 4); return r; }
static inline int shfl_idx(int v, int src) { const uint32_t* s = exchange((uint32_t)v); return (int)s[src & 63]; }
static inline int uniform(int v) { return v; }
static inline float uniformf(float v) { return v; }
static inline uint64_t ballot(bool p) { const uint32_t* s = exchange(p ? 1u : 0u); uint64_t m = 0; for (int i = 0; i < 64; i++) m |= (uint64_t)(s[i] & 1) << i; return m; }
static inline float wave_sum(float v) { uint32_t u; memcpy(&u, &v, 4); const uint32_t* s = exchange(u); float r = 0; for (int i = 0; i < 64; i++) { float t; memcpy(&t, &s[i], 4); r += t; } return r; }
static inline float wave_min(float v) { uint32_t u; memcpy(&u, &v, 4); const uint32_t* s = exchange(u); float r = INFINITY; for (int i = 0; i < 64; i++) { float t; memcpy(&t, &s[i], 4); r = std::min(r, t); } return r; }
static inline float wave_max(float v) { uint32_t u; memcpy(&u, &v, 4); const uint32_t* s = exchange(u); float r = -INFINITY; for (int i = 0; i < 64; i++) { float t; memcpy(&t, &s[i], 4); r = std::max(r, t); } return r; }
static inline float quad_sum(float v) { uint32_t u; memcpy(&u, &v, 4); const uint32_t* s = exchange(u); const int q = lane_id() & ~3; float r = 0; for (int i = 0; i < 4; i++) { float t; memcpy(&t, &s[q + i], 4); r += t; } return r; }
static inline int wave_incl_scan(int v) { const uint32_t* s = exchange((uint32_t)v); int r = 0; for (int i = 0; i <= lane_id(); i++) r += (int)s[i]; return r; }
static inline int popc64(uint64_t m) { return __builtin_popcountll(m); }
static inline int ffs64(uint64_t m) { return __builtin_ctzll(m); }
static inline void opaque(int&) {}
static inline void opaque_s(int&) {}
template <class T> static inline void pin(T&) {}
template <class T> static inline const T* opaque_ptr(const T* p) { return p; }
static inline int opaque_lane(int l) { return l; }
template <class T> static inline T* gptr(T* p) { return p; }
template <class T> static inline const T* mptr(const T* p) { return p; }
template <bool PUB, class T> static inline T ldv(const T* p) { return *p; }
static inline int ld_pub(const int32_t* p) { return *p; }
static inline float ld_pub(const float* p) { return *p; }
static inline void st_pub(int32_t* p, int v) { *p = v; }
static inline void st_pub(float* p, float v) { *p = v; }
static inline int add_pub(int32_t* p, int v) { const int o = *p; *p = o + v; return o; }
inline int g_emu_cas_ok = 0; /* successful compare-and-swaps (the pair exchange: reservations + claims) */
static inline bool cas_pub(int32_t* p, int expect, int v) { if (*p != expect) return false; *p = v; g_emu_cas_ok++; return true; }
static inline void publish_fence() {}
static inline void nap() {}
static inline void sched_fence() {}
static inline long long cycles() { return 0; }
static inline long long wall_clock64() { static long long t = 0; return t += 100; } /* a microsecond per look: the pair exchange's waits (gq_exchange.h) end */
static inline void wave_priority(int) {}
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
#define __builtin_amdgcn_s_getreg(x) 0
static inline float fast_rcp(float x) { return 1.0f / x; }
static inline float fast_rsqrt(float x) { return 1.0f / std::sqrt(x); }
static inline float fdiv(float a, float b) { return a / b; }
static inline float fast_sqrt(float x) { return std::sqrt(x); }
static inline float fast_cos_turns(float x) { return std::cos(6.283185307179586f * x); }
static inline float fast_pow_ratio(float a, float p, float b, float q) { return std::exp2(p * std::log2(a) - q * std::log2(b)); }
static inline float med3(float x, float lo, float hi) { return std::min(std::max(x, lo), hi); }
}  // namespace gq
