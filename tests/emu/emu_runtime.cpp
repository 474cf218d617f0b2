// TEST INFRASTRUCTURE ONLY -- fiber scheduler behind tests/emu/include/hip/hip_runtime.h.
#include <hip/hip_runtime.h>

uint3_emu threadIdx, blockIdx;
dim3 blockDim, gridDim;
alignas(64) unsigned char aa_smem[160 * 1024];

namespace emu {
static State g_state;
State& st() { return g_state; }

static constexpr size_t STACK_BYTES = 256 * 1024;

static void set_thread_idx(int t) {
  threadIdx.x = t % blockDim.x;
  threadIdx.y = (t / blockDim.x) % blockDim.y;
  threadIdx.z = t / (blockDim.x * blockDim.y);
}

static void fiber_main() {
  State& s = st();
  s.body();
  s.fibers[s.cur].done = true;
  swapcontext(&s.fibers[s.cur].ctx, &s.sched);
}

static void yield_with(int kind) {
  State& s = st();
  Fiber& f = s.fibers[s.cur];
  f.wait_kind = kind;
  swapcontext(&f.ctx, &s.sched);
}
void yield_block_barrier() { yield_with(1); }
void wave_rendezvous() { yield_with(2); }

static void run_block() {
  State& s = st();
  int n = s.nthreads;
  for (int t = 0; t < n; ++t) {
    Fiber& f = s.fibers[t];
    f.done = false;
    f.wait_kind = 0;
    getcontext(&f.ctx);
    if (!f.stack) f.stack = (char*)malloc(STACK_BYTES);
    f.ctx.uc_stack.ss_sp = f.stack;
    f.ctx.uc_stack.ss_size = STACK_BYTES;
    f.ctx.uc_link = &s.sched;
    makecontext(&f.ctx, fiber_main, 0);
  }
  // Round-robin: run every runnable fiber until it blocks; release a barrier when all live fibers
  // of its scope (block / wave) wait on it.
  for (;;) {
    bool progressed = false;
    int live = 0;
    for (int t = 0; t < n; ++t) {
      Fiber& f = s.fibers[t];
      if (f.done) continue;
      ++live;
      if (f.wait_kind == 0) {
        s.cur = t;
        set_thread_idx(t);
        swapcontext(&s.sched, &f.ctx);
        progressed = true;
      }
    }
    if (live == 0) break;
    // wave rendezvous release
    int nw = (n + WAVE - 1) / WAVE;
    for (int w = 0; w < nw; ++w) {
      int lo = w * WAVE, hi = lo + WAVE < n ? lo + WAVE : n;
      int waiting = 0, alive = 0;
      for (int t = lo; t < hi; ++t) {
        if (s.fibers[t].done) continue;
        ++alive;
        if (s.fibers[t].wait_kind == 2) ++waiting;
      }
      if (alive > 0 && waiting == alive) {
        for (int t = lo; t < hi; ++t)
          if (!s.fibers[t].done) s.fibers[t].wait_kind = 0;
        progressed = true;
      }
    }
    // block barrier release
    int waiting = 0, alive = 0;
    for (int t = 0; t < n; ++t) {
      if (s.fibers[t].done) continue;
      ++alive;
      if (s.fibers[t].wait_kind == 1) ++waiting;
    }
    if (alive > 0 && waiting == alive) {
      for (int t = 0; t < n; ++t)
        if (!s.fibers[t].done) s.fibers[t].wait_kind = 0;
      progressed = true;
    }
    if (!progressed) {
      fprintf(stderr, "[hip emu] deadlock: divergent barrier / wave collective\n");
      abort();
    }
  }
}

void launch(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body) {
  State& s = st();
  if (smem > sizeof(aa_smem)) {
    fprintf(stderr, "[hip emu] dynamic LDS %zu > 160 KiB\n", smem);
    abort();
  }
  s.nthreads = block.x * block.y * block.z;
  if ((int)s.fibers.size() < s.nthreads) s.fibers.resize(s.nthreads);
  s.xchg.assign(size_t((s.nthreads + WAVE - 1) / WAVE) * WAVE * 4, 0.0);
  s.body = body;
  blockDim = block;
  gridDim = grid;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        blockIdx.x = bx;
        blockIdx.y = by;
        blockIdx.z = bz;
        run_block();
      }
}
}  // namespace emu
