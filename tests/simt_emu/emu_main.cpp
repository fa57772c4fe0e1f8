/* TEST INFRASTRUCTURE - coroutine scheduler of the host SIMT emulator (see gq_device.h in this directory). */
#include <ucontext.h>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include "gq_device.h"

EmuDim3 threadIdx, blockIdx, blockDim = {64, 1, 1}, gridDim = {1, 1, 1};
uint32_t emu_slot[2][GQ_WAVE];
int emu_phase[GQ_WAVE];

static ucontext_t g_main, g_ctx[GQ_WAVE];
static bool g_done[GQ_WAVE];
static int g_cur;
static std::function<void()> g_body;
static char* g_stacks;
static const size_t STACK = 1 << 20;

void emu_yield() {
  int me = g_cur;
  swapcontext(&g_ctx[me], &g_main);
  threadIdx.x = (unsigned)me;
}
static void trampoline() {
  g_body();
  g_done[g_cur] = true;
  swapcontext(&g_ctx[g_cur], &g_main);
}
/* run `body` as one 64-lane wavefront for block `block` */
void emu_run_wave(unsigned block, unsigned nblocks, const std::function<void()>& body) {
  if (!g_stacks) g_stacks = (char*)malloc(STACK * GQ_WAVE);
  g_body = body;
  blockIdx.x = block; gridDim.x = nblocks;
  for (int l = 0; l < GQ_WAVE; l++) {
    getcontext(&g_ctx[l]);
    g_ctx[l].uc_stack.ss_sp = g_stacks + STACK * l;
    g_ctx[l].uc_stack.ss_size = STACK;
    g_ctx[l].uc_link = &g_main;
    makecontext(&g_ctx[l], trampoline, 0);
    g_done[l] = false; emu_phase[l] = 0;
  }
  for (;;) {
    int alive = 0;
    for (int l = 0; l < GQ_WAVE; l++) {
      if (g_done[l]) continue;
      alive++;
      g_cur = l; threadIdx.x = (unsigned)l;
      swapcontext(&g_main, &g_ctx[l]);
    }
    if (!alive) break;
  }
}
