/* Convex pair exchange: the convex narrow phase (gq_convex.h) of a launch, shared between its wavefronts.
 *
 * Why.  One env per wavefront, and a launch lasts as long as its slowest wavefront.  The convex routine is a dependent chain - 12 us for
 * the median hull pair that needs it, 40 - 80 us for a penetrating one whose polytope takes 10 - 20 expansions - and its load is as uneven
 * as a load can be: of 4096 benchmark envs six in ten have no pair that needs it, a handful have ten to fourteen (a robot folded onto
 * itself); tools/convex_census.py, tools/wave_timeline.py.  Run where they arise, those few envs kept 4000 finished wavefronts waiting for
 * 300 us.  A pair is a self-contained unit of work - two posed shapes and a margin in, a distance, a normal and a point out, 41 words and
 * 8 - so the wavefront that owns several hands all but one of them to whoever has time:
 *
 *   owner    keeps the first of its pairs and PUBLISHES the others, each into a slot of the batch's table that the pair's lane reserves
 *            (one compare-and-swap per pair, all in flight together) - at once, before it knows which of them are the expensive ones
 *            (most separate after a query or two): a pair published late finds nobody; works on its own pair; then watches its slots - a
 *            pair nobody has taken yet it takes back - and collects the results;
 *   helpers  an env without convex work of its own has tens of microseconds to spare before the entangled envs are through: at the
 *            convex block of its step it LINGERS for 25 us - if a pair was published into its window within the last 20 ms (robots stay
 *            entangled for many steps; the window's time word is fetched at the head of the stage, behind the pair cull) - looks at ITS window of the table, 124 slots, two words per lane, every
 *            microsecond or so, and computes what is READY there.  Pairs are taken within a microsecond or two of their publication
 *            (measured: READY 25 us after the owner's start, first claim after 27).
 *
 * The result of a pair does not depend on who computes it (same code, same inputs), so the contact list is the one the owner would have
 * built alone: bit-identical - the parity tests do not know the exchange exists, tests/test_gpu_parity.py compares on and off.
 *
 * No shared counter.  (The first version had a head / tail queue: 4000 wavefronts leaving within a few microseconds, each with a
 * compare-and-swap on the tail word while items were left, cost 65 MILLIseconds per launch - device-scope atomics on ONE address are
 * served one after the other, tens of nanoseconds each.)  A slot is reserved, claimed and freed by a compare-and-swap / store on its own
 * state word; pairs are spread over the table by a hash of (wavefront, lane), helpers over it by their wavefront index: an atomic is
 * only ever issued on a word that holds a pair, by the wavefronts that share its window.
 *
 * Slot life: FREE -owner: CAS-> RESERVED -owner: item written-> READY -helper or owner: CAS-> CLAIMED -result written-> DONE -owner: result
 * read-> FREE.  An owner trusts only slots it reserved itself in this step, so launches that overlap on several streams (gq_rollout's shards,
 * gq_step_range) share the table safely, a persistent rollout reuses its slots step after step, and nothing has to be rewound between
 * launches.  Progress: a pair is either READY - its owner will take it back - or in the hands of a running wavefront, whose routine is
 * bounded; an owner that has waited GQ_XQ_OWNER_TICKS for a CLAIMED pair computes it itself and leaves the slot behind (never reused: the
 * late result cannot land on somebody else's pair) - a safety net that no run has needed.  No launch-wide barrier, no assumption that the
 * launch is resident at once.  (A second visit of every wavefront at the END of its step, tried first, changed nothing once the
 * lingering was in: removed.)
 *
 * What it buys and what is left (MI355X, 4096 mini_cheetah envs, steady state): the launch 375 -> 147 us.  What remains is one chain:
 * the owner's pairs are READY after 25 - 37 us, the launch's most expensive pair takes 77 us whoever computes it, the owner's solver and
 * epilogue 25 - 30 us.  Only a faster routine shortens it further.
 *
 * Memory.  Table words travel between wavefronts of different XCDs, i.e. different L2s: every access is an agent-scope relaxed atomic
 * (ld_pub / st_pub / cas_pub, gq_device.h - sc1: served by / written through to the device-coherent level), ordered by publish_fence()
 * in front of the state word's store and by the consumer's dependence on that word.  No device-scope acquire fence (it empties the XCD's L2).
 *
 * Layout (int32 words): slots state words (a power of two >= 256; every 32nd word is no slot - a window's word 31 holds the time of the last
 * publication into it), then slots items of GQ_XQ_ITEM words: shape A, shape B (20 + 20, CvxShape), margin, ..., result (hit, dist, normal,
 * point) at GQ_XQ_RES.  (Constants: gq_step_kernel.h - the host allocates by them.) */
#pragma once
#include "gq_convex.h"

namespace gq {

#define GQ_XQ_OWNER_TICKS 200000 /* 100 MHz ticks an owner waits for a CLAIMED pair before it computes it itself: 2 ms */
enum { XQ_FREE = 0, XQ_RESERVED = 1, XQ_READY = 2, XQ_CLAIMED = 3, XQ_DONE = 4 };

#define GQ_XQ_LINGER_TICKS 2500  /* 100 MHz ticks an env without convex work waits at the convex block for pairs to be published: 25 us (an owner needs ten to get there - candidates, mid phase, reservation, the pairs, the fence - and wavefronts start up to 10 us apart) */
#define GQ_XQ_HOT_UNITS 125000   /* ... if a pair was published into its window within this many 160 ns units: 20 ms (robots stay entangled for many steps; a policy between two steps takes its milliseconds) */
struct Xq { int32_t* q; int slots; }; /* q = nullptr: no exchange */
__device__ __forceinline__ int32_t* xq_state(const Xq& x, int slot) { return x.q + slot; }
__device__ __forceinline__ int32_t* xq_item(const Xq& x, int slot) { return x.q + x.slots + (size_t)slot * GQ_XQ_ITEM; }

/* a helper's window: 128 consecutive state words (one load of two words per lane), 124 slots.  Windows do not overlap: slots / 128 of them,
 * each watched by the wavefronts whose index is congruent - on the benchmark 64 windows of 124 slots, 64 wavefronts (45 of them without
 * convex work of their own) and some 35 pairs per launch each: wide enough that a pair rarely waits while a helper of its window idles. */
__device__ __forceinline__ int xq_window(const Xq& x, int widx) { return (widx & ((x.slots >> 7) - 1)) << 7; }
/* look at the window: a READY slot for this wavefront to try (index into the table; the wavefronts that share a window start on
 * different pairs), or -1 */
__device__ __forceinline__ int xq_scan(const Xq& x, int widx) {
  const int lane = lane_id();
  const int base = xq_window(x, widx);
  const int s0 = ld_pub(xq_state(x, base + lane)), s1 = ld_pub(xq_state(x, base + 64 + lane));
  const uint64_t r0 = ballot((lane & 31) != 31 && s0 == XQ_READY), r1 = ballot((lane & 31) != 31 && s1 == XQ_READY);
  const int n0 = popc64(r0), n = n0 + popc64(r1);
  if (n == 0) return -1;
  int k = (widx / (x.slots >> 7)) % n;
  uint64_t r = r0; int off = 0;
  if (k >= n0) { r = r1; off = 64; k -= n0; }
  for (; k > 0; k--) r &= r - 1;
  return base + off + ffs64(r);
}
__device__ __forceinline__ int xq_time_units(long long ticks) { return (int)((ticks >> 4) & 0x3fffffff) | 1; } /* (never 0: the word's initial value) */
/* a window's time word (its word 31: every 32nd state word is no slot): when a pair was last published into the window, written by the
 * publishing lane.  (ONE such word for the whole table, read by every wavefront, was tried: 4096 device-coherent loads of one address
 * delayed the stage by 14 us - and a plain load never sees the update, each XCD's L2 keeps the line it fetched first.) */
__device__ __forceinline__ int32_t* xq_time_word(const Xq& x, int slot) { return x.q + ((slot & ~127) | 31); }
/* ... and its word 63: when an env without convex work last passed by the convex block - owners publish only while such envs are around.
 * (Rewritten once a millisecond at most: hlp = the value the caller fetched.) */
__device__ __forceinline__ void xq_mark_helper(const Xq& x, int widx, int hlp) {
  const long long now = wall_clock64();
  if (lane_id() == 0 && (hlp == 0 || ((xq_time_units(now) - hlp) & 0x3fffffff) > 6250)) st_pub(x.q + xq_window(x, widx) + 63, xq_time_units(now));
}
__device__ __forceinline__ void xq_mark_active(const Xq& x, int slot) { st_pub(xq_time_word(x, slot), xq_time_units(wall_clock64())); }
__device__ __forceinline__ bool xq_is_hot(int word, long long now_ticks) { return word != 0 && ((xq_time_units(now_ticks) - word) & 0x3fffffff) < GQ_XQ_HOT_UNITS; }


/* (per lane) reserve a slot for the pair of lane `lane` of wavefront `widx`: its index, or -1 after four occupied candidates */
__device__ __forceinline__ int xq_reserve(const Xq& x, int widx, int lane) {
  const uint32_t h = (uint32_t)widx * 2654435761u + (uint32_t)lane * 40503u;
  for (int t = 0; t < 4; t++) {
    int slot = (int)(((h >> 9) + (uint32_t)t * 977u) & (uint32_t)(x.slots - 1));
    if ((slot & 31) == 31) slot--; /* (no slot: the windows' time words live there) */
    if (cas_pub(xq_state(x, slot), XQ_FREE, XQ_RESERVED)) return slot;
  }
  return -1;
}
/* (wave-uniform) write the pair in shp (LDS, 2 x GQ_CVX_SHAPE_WORDS) + margin into the reserved slot; xq_ready follows */
__device__ __forceinline__ void xq_put(const Xq& x, int slot, LdsCF shp, float margin) {
  int32_t* it = xq_item(x, slot);
  const int lane = lane_id();
  if (lane < 2 * GQ_CVX_SHAPE_WORDS) st_pub(it + lane, ((LdsCI)shp)[lane]);
  if (lane == GQ_XQ_MARGIN) st_pub(reinterpret_cast<float*>(it) + GQ_XQ_MARGIN, margin);
}
/* (per lane, after a publish_fence) the pair is there */
__device__ __forceinline__ void xq_ready(const Xq& x, int slot) { st_pub(xq_state(x, slot), XQ_READY); }
/* (wave-uniform) take a READY pair: false when somebody else was faster */
__device__ __forceinline__ bool xq_claim(const Xq& x, int slot) {
  int ok = 0;
  if (lane_id() == 0) ok = cas_pub(xq_state(x, slot), XQ_READY, XQ_CLAIMED) ? 1 : 0;
  return bcast(ok, 0) != 0;
}
/* (wave-uniform) fetch a claimed pair into shp */
__device__ __forceinline__ void xq_get(const Xq& x, int slot, LdsF shp, float& margin) {
  const int32_t* it = xq_item(x, slot);
  const int lane = lane_id();
  int w = 0;
  if (lane < 2 * GQ_CVX_SHAPE_WORDS) w = ld_pub(it + lane);
  margin = uniformf(ld_pub(reinterpret_cast<const float*>(it) + GQ_XQ_MARGIN));
  if (lane < 2 * GQ_CVX_SHAPE_WORDS) ((LdsI)shp)[lane] = w;
}
/* (wave-uniform) store the result of a claimed pair (out: the 7 result words cvx_pair_wave left behind the shapes) and mark it DONE */
__device__ __forceinline__ void xq_done(const Xq& x, int slot, bool hit, LdsCF out) {
  int32_t* it = xq_item(x, slot);
  const int lane = lane_id();
  if (lane < 8) {
    const int w = lane == 0 ? (hit ? 1 : 0) : ((LdsCI)out)[lane - 1];
    st_pub(it + GQ_XQ_RES + lane, w);
  }
  publish_fence();
  if (lane == 0) st_pub(xq_state(x, slot), XQ_DONE);
}

}  // namespace gq
