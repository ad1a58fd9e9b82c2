// TEST INFRASTRUCTURE ONLY -- fiber scheduler behind tests/emu/hip/hip_runtime.h.
#include <hip/hip_runtime.h>

#include <chrono>
#include <mutex>

namespace hipemu {
thread_local Ctx* cur = nullptr;

namespace {
constexpr size_t kStack = 96 * 1024;

struct Worker {
  std::vector<ucontext_t> fibers;
  std::vector<Ctx> ctx;
  std::vector<char> stacks;
  std::vector<char> smem;
  ucontext_t sched;
};

thread_local const std::function<void()>* g_body = nullptr;

void trampoline() {
  (*g_body)();
  cur->done = true;
  swapcontext(cur->self, cur->sched);
}

void run_block(Worker& w, dim3 grid, dim3 block, size_t shmem, unsigned bx, unsigned by,
               const std::function<void()>& body) {
  const unsigned nt = block.x * block.y * block.z;
  if (w.fibers.size() < nt) {
    w.fibers.resize(nt);
    w.ctx.resize(nt);
    w.stacks.resize(size_t(nt) * kStack);
  }
  if (w.smem.size() < shmem + 64) w.smem.resize(shmem + 64);
  char* smem = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(w.smem.data()) + 63) & ~uintptr_t(63));
  g_body = &body;
  for (unsigned t = 0; t < nt; ++t) {
    Ctx& c = w.ctx[t];
    c.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
    c.bid = dim3(bx, by, 0);
    c.bdim = block;
    c.gdim = grid;
    c.smem = smem;
    c.self = &w.fibers[t];
    c.sched = &w.sched;
    c.done = false;
    getcontext(&w.fibers[t]);
    w.fibers[t].uc_stack.ss_sp = w.stacks.data() + size_t(t) * kStack;
    w.fibers[t].uc_stack.ss_size = kStack;
    w.fibers[t].uc_link = &w.sched;
    makecontext(&w.fibers[t], trampoline, 0);
  }
  unsigned alive = nt;
  while (alive) {
    unsigned finished = 0;
    for (unsigned t = 0; t < nt; ++t) {
      if (w.ctx[t].done) continue;
      cur = &w.ctx[t];
      swapcontext(&w.sched, &w.fibers[t]);
      if (w.ctx[t].done) ++finished;
    }
    alive -= finished;
    if (finished && alive) {
      fprintf(stderr, "hipemu: %u threads left a block early while %u still run (barrier divergence)\n",
              finished, alive);
      abort();
    }
  }
  cur = nullptr;
}
}  // namespace

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
  const unsigned nblocks = grid.x * grid.y;
  unsigned nthreads = std::thread::hardware_concurrency();
  if (nthreads == 0) nthreads = 4;
  if (nthreads > nblocks) nthreads = nblocks;
  std::atomic<unsigned> next{0};
  auto work = [&]() {
    Worker w;
    for (;;) {
      unsigned b = next.fetch_add(1);
      if (b >= nblocks) break;
      run_block(w, grid, block, shmem, b % grid.x, b / grid.x, body);
    }
  };
  if (nthreads <= 1) {
    work();
  } else {
    std::vector<std::thread> pool;
    for (unsigned i = 0; i < nthreads; ++i) pool.emplace_back(work);
    for (auto& t : pool) t.join();
  }
}
}  // namespace hipemu

double hipemu_now_ms() {
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}
