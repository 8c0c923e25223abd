// hipemu scheduler — see hipemu.h.  TEST INFRASTRUCTURE ONLY.
#include "hipemu.h"

#include <ucontext.h>

#include <vector>

namespace hipemu {

dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;

namespace {
constexpr size_t kStackBytes = 256 * 1024;
constexpr int kMaxThreads = 1024;
constexpr int kMaxWaves = kMaxThreads / 64;

struct Fiber {
  ucontext_t ctx;
  char* stack = nullptr;
  bool done = false;
  dim3 tid;
  int linear = 0;
};

struct Barrier {
  int count = 0;
  unsigned gen = 0;
};

Fiber g_fibers[kMaxThreads];
ucontext_t g_sched;
int g_nthreads = 0;
int g_live = 0;      // threads of the current block that have not returned: what a block barrier waits for (as s_barrier
                     // counts the waves that are still alive — kernels may retire whole waves before a barrier)
int g_current = -1;
unsigned long g_progress = 0;  // bumped whenever any barrier releases or a fiber ends
Barrier g_block_bar;
Barrier g_wave_bar[kMaxWaves];
uint64_t g_slots[2][kMaxWaves][64];
const std::function<void()>* g_body = nullptr;

void yield_to_scheduler() {
  Fiber& f = g_fibers[g_current];
  swapcontext(&f.ctx, &g_sched);
}

void thread_retired();
void trampoline() {
  (*g_body)();
  g_fibers[g_current].done = true;
  ++g_progress;
  thread_retired();
  yield_to_scheduler();
}

void thread_retired() {
  --g_live;
  if (g_live > 0 && g_block_bar.count == g_live) {      // everyone still alive is already parked at the block barrier
    g_block_bar.count = 0;
    ++g_block_bar.gen;
    ++g_progress;
  }
}

void wait_on(Barrier& b, int participants) {
  unsigned gen = b.gen;
  if (++b.count == participants) {
    b.count = 0;
    ++b.gen;
    ++g_progress;
    return;
  }
  while (b.gen == gen) yield_to_scheduler();
}
}  // namespace

int lane_id() { return g_fibers[g_current].linear & 63; }
int wave_id() { return g_fibers[g_current].linear >> 6; }
int wave_width() {
  int w = wave_id();
  int rem = g_nthreads - w * 64;
  return rem >= 64 ? 64 : rem;
}
uint64_t* wave_slots(int which) { return g_slots[which][wave_id()]; }
void block_barrier() { wait_on(g_block_bar, g_live); }
void wave_barrier() { wait_on(g_wave_bar[wave_id()], wave_width()); }

void launch(const std::function<void()>& body, dim3 grid, dim3 block) {
  const int nthreads = (int)(block.x * block.y * block.z);
  if (nthreads <= 0 || nthreads > kMaxThreads) {
    std::fprintf(stderr, "hipemu: bad block size %d\n", nthreads);
    std::abort();
  }
  g_body = &body;
  g_blockDim = block;
  g_gridDim = grid;
  g_nthreads = nthreads;
  for (int i = 0; i < nthreads; ++i)
    if (!g_fibers[i].stack) g_fibers[i].stack = (char*)std::malloc(kStackBytes);

  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        g_block_bar = Barrier();
        for (auto& wb : g_wave_bar) wb = Barrier();
        int li = 0;
        for (unsigned tz = 0; tz < block.z; ++tz)
          for (unsigned ty = 0; ty < block.y; ++ty)
            for (unsigned tx = 0; tx < block.x; ++tx, ++li) {
              Fiber& f = g_fibers[li];
              f.done = false;
              f.tid = dim3(tx, ty, tz);
              f.linear = li;
              getcontext(&f.ctx);
              f.ctx.uc_stack.ss_sp = f.stack;
              f.ctx.uc_stack.ss_size = kStackBytes;
              f.ctx.uc_link = &g_sched;
              makecontext(&f.ctx, (void (*)())trampoline, 0);
            }
        g_live = nthreads;
        int remaining = nthreads;
        while (remaining > 0) {
          unsigned long before = g_progress;
          remaining = 0;
          for (int i = 0; i < nthreads; ++i) {
            Fiber& f = g_fibers[i];
            if (f.done) continue;
            g_current = i;
            g_threadIdx = f.tid;
            g_blockIdx = dim3(bx, by, bz);
            swapcontext(&g_sched, &f.ctx);
            if (!f.done) ++remaining;
          }
          if (remaining > 0 && g_progress == before) {
            std::fprintf(stderr,
                         "hipemu: DEADLOCK in block (%u,%u,%u): %d thread(s) parked at a barrier the rest "
                         "never reached (divergent __syncthreads / wave collective)\n",
                         bx, by, bz, remaining);
            std::abort();
          }
        }
      }
  g_current = -1;
  g_body = nullptr;
}

}  // namespace hipemu
